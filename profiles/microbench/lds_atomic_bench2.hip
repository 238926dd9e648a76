// LDS atomic throughput, address arithmetic kept out of the loop (immediate
// offsets), 5 workgroups of 256 threads per CU like the z-buffer tile kernel.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_bench2.hip -o lds_atomic_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;

// PATTERN 0: lane l -> slot l (64 consecutive 8-byte slots)
// PATTERN 1: quarter q -> column q*7, rows ly        (4 lanes per bank pair)
// PATTERN 2: quarter q -> column q*7, rows ly + 16*(q&1)  (2 lanes per bank pair)
// MODE 4: u64 atomicMin with only `active` lanes per wave enabled
// MODE 5: b64 read + compare, atomicMin only where the key is smaller (keys decrease every
//         `period` iterations so that roughly 1/period of the tests pass)
template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k(int iters, u64 *out, int active = 64, int period = 1)
{
    __shared__ u64 zb[4096];
    for (int p = threadIdx.x; p < 4096; p += 256) zb[p] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, ly = lane & 15;
    int slot = lane;
    if (PATTERN == 1) slot = (q * 7) * 64 + ly;
    if (PATTERN == 2) slot = (q * 7) * 64 + ly + 16 * (q & 1);
    u64 *p = zb + slot;
    unsigned *p32 = (unsigned *)zb + slot;
    u64 key = ((u64)(1000 + blockIdx.x) << 32) | threadIdx.x;
    u64 acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) atomicMin(p + u * 64, key);
            if (MODE == 1) atomicMin(p32 + u * 64, (unsigned)key);
            if (MODE == 2) p[u * 64] = key;
            if (MODE == 3) acc += p[u * 64];
            if (MODE == 4) { if (lane < active) atomicMin(p + u * 64, key); }
            if (MODE == 5) { const u64 cur = *(volatile u64 *)(p + u * 64); if (key < cur) atomicMin(p + u * 64, key); }
        }
        if (MODE == 5) { if (it % period == period - 1) key -= 1; } else key -= 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = zb[5] + acc;
}

template <int MODE, int PATTERN> void run(const char *name)
{
    const int blocks = 256 * 5 * 2, iters = 1000;
    u64 *out; (void)hipMalloc(&out, blocks * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, PATTERN><<<blocks, 256>>>(iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<MODE, PATTERN><<<blocks, 256>>>(iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 8;      // wave-level LDS instructions
    printf("%-22s pattern %d  %8.3f ms  %.2f cycles/wave-instr/CU @2.4GHz  %.1f lanes/clk/CU\n", name,
           PATTERN, ms, ms * 1e-3 * 2.4e9 * 256 / winstr, winstr * 64 / (ms * 1e-3 * 2.4e9 * 256));
    (void)hipFree(out);
}

template <int MODE> void run2(const char *name, int active, int period)
{
    const int blocks = 256 * 5 * 2, iters = 1000;
    u64 *out; (void)hipMalloc(&out, blocks * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, 2><<<blocks, 256>>>(iters, out, active, period);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<MODE, 2><<<blocks, 256>>>(iters, out, active, period);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 8;
    printf("%-22s active %2d period %2d  %8.3f ms  %.2f cycles/wave-instr/CU @2.4GHz\n", name, active, period,
           ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
    (void)hipFree(out);
}

int main()
{
    for (int act : {64, 48, 32, 16, 8}) run2<4>("u64 atomic masked", act, 1);
    for (int per : {1, 2, 4, 8, 1000}) run2<5>("read-test-atomic", 64, per);
    run<0, 0>("u64 atomicMin"); run<0, 1>("u64 atomicMin"); run<0, 2>("u64 atomicMin");
    run<1, 0>("u32 atomicMin"); run<1, 1>("u32 atomicMin"); run<1, 2>("u32 atomicMin");
    run<2, 0>("b64 write"); run<2, 1>("b64 write"); run<2, 2>("b64 write");
    run<3, 0>("b64 read"); run<3, 1>("b64 read"); run<3, 2>("b64 read");
    return 0;
}
