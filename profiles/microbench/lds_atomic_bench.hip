// Microbenchmark: LDS atomic throughput on gfx950 for the z-buffer tile kernel.
// Each 256-thread workgroup owns a 32 KB LDS tile and issues ITER wave-wide
// atomics per wave with a given lane->address pattern.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_bench.hip -o lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int hlog, u64 *out)
{
    __shared__ u64 zb[4096];
    unsigned *zb32 = (unsigned *)zb;
    for (int p = threadIdx.x; p < 4096; p += 256) zb[p] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ly = lane & ((1 << hlog) - 1), lx = lane >> hlog;
    u64 key = ((u64)(1000 + blockIdx.x) << 32) | threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int x = (it * 7 + wave * 13 + lx) & 63;
        const int y = (it * 3 + ly) & 63;
        if (MODE == 0) atomicMin(&zb[x * 64 + y], key + it);                 // u64 no return
        if (MODE == 1) atomicMin(&zb32[x * 64 + y], (unsigned)key + it);     // u32 no return
        if (MODE == 2) { if (key + it < zb[x * 64 + y]) atomicMin(&zb[x * 64 + y], key + it); }  // read-test-atomic
        if (MODE == 3) { u64 o = zb[x * 64 + y]; key += o & 1; }             // plain b64 read
        if (MODE == 4) { u64 o = atomicMin(&zb[x * 64 + y], key + it); key += o & 1; }  // u64 with return
        if (MODE == 5) atomicMin(&zb[x * 65 + y], key + it);                 // padded rows (conflict pattern)
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = zb[5] + key;
}

template <int MODE> void run(const char *name, int hlog)
{
    const int blocks = 256 * 5 * 4, iters = 2000;
    u64 *out; hipMalloc(&out, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(iters, hlog, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(iters, hlog, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters;
    printf("%-28s hlog=%d  %8.3f ms  %8.2f Gops/s  %.2f lanes/clk/CU @2.4GHz\n", name, hlog, ms,
           ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main()
{
    for (int hlog : {6, 4, 2}) {
        run<0>("u64 atomicMin noret", hlog);
        run<1>("u32 atomicMin noret", hlog);
        run<2>("u64 read-test-atomic", hlog);
        run<3>("b64 plain read", hlog);
        run<4>("u64 atomicMin ret", hlog);
        run<5>("u64 atomicMin padded rows", hlog);
    }
    return 0;
}
