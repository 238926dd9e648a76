// Issue cost of LDS instructions on gfx950, in cycles per wave-instruction per
// CU, for the instruction kinds a tiled z-buffer can be built from: 64-bit and
// 32-bit atomic minimum (with and without returned value), plain reads and
// writes of 4 / 8 / 16 bytes per lane, with 64 or 16 active lanes per wave.
// 512-thread workgroups, 4 per CU (the z-buffer kernel's shape); every lane
// works on its own slot (no bank conflicts beyond the data width).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

enum { MIN64, MIN32, MIN32_RTN, MIN64_RTN, RD32, RD64, RD128, WR64, WR128, NKIND };
static const char *kind_name[NKIND] = {"ds_min_u64", "ds_min_u32", "ds_min_rtn_u32", "ds_min_rtn_u64",
                                       "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_write_b64", "ds_write_b128"};

template <int KIND, int ACTIVE>
__global__ __launch_bounds__(512) void k(int iters, u64 *out)
{
    __shared__ __align__(16) u64 zb[4608];           // 36 KB, as the z-buffer tile
    for (int p = threadIdx.x; p < 4608; p += 512) zb[p] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = (lane & 63) < ACTIVE;
    // lane-private slots, 8 unrolled positions 64 * 16 bytes apart at most
    char *base = (char *)zb + lane * (KIND == RD128 || KIND == WR128 ? 16 : KIND == MIN32 || KIND == MIN32_RTN || KIND == RD32 ? 4 : 8);
    const int stride = KIND == RD128 || KIND == WR128 ? 1024 : KIND == MIN32 || KIND == MIN32_RTN || KIND == RD32 ? 256 : 512;
    u64 key = ((u64)(1000000 + blockIdx.x) << 32) | threadIdx.x;
    u64 acc = 0;
    const unsigned a0 = (unsigned)(size_t)base;      // LDS byte address of the lane's first slot
    if (on)
        for (int it = 0; it < iters; ++it) {
            unsigned r32[8]; u64 r64[8]; v4u r128[8];
            const v4u w128 = {(unsigned)key, 1u, 2u, 3u};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                char *p = base + u * stride;
                const unsigned a = a0 + u * stride;
                if (KIND == MIN64) atomicMin((u64 *)p, key);
                else if (KIND == MIN32) atomicMin((unsigned *)p, (unsigned)key);
                else if (KIND == MIN32_RTN) acc += atomicMin((unsigned *)p, (unsigned)key);
                else if (KIND == MIN64_RTN) acc += atomicMin((u64 *)p, key);
                else if (KIND == RD32) asm volatile("ds_read_b32 %0, %1" : "=v"(r32[u]) : "v"(a));
                else if (KIND == RD64) asm volatile("ds_read_b64 %0, %1" : "=v"(r64[u]) : "v"(a));
                else if (KIND == RD128) asm volatile("ds_read_b128 %0, %1" : "=v"(r128[u]) : "v"(a));
                else if (KIND == WR64) asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(key) : "memory");
                else if (KIND == WR128) asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(w128) : "memory");
            }
            if (KIND >= RD32) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (KIND == RD32) acc += r32[u];
                    else if (KIND == RD64) acc += r64[u];
                    else if (KIND == RD128) acc += r128[u].x + r128[u].w;
                }
            }
            key -= 1;
        }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = zb[5] + acc;
}

template <int KIND, int ACTIVE> void run()
{
    const int blocks = 256 * 4, iters = 1000;
    u64 *out; (void)hipMalloc(&out, blocks * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<KIND, ACTIVE><<<blocks, 512>>>(iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<KIND, ACTIVE><<<blocks, 512>>>(iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 8 * iters * 8;     // 8 waves, 8 unrolled
    printf("%-16s active lanes %2d  %8.3f ms  %6.2f cycles/wave-instr/CU @2.4GHz\n",
           kind_name[KIND], ACTIVE, ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
    (void)hipFree(out);
}

int main()
{
    run<MIN64, 64>(); run<MIN64, 16>();
    run<MIN32, 64>(); run<MIN32, 16>();
    run<MIN32_RTN, 64>(); run<MIN64_RTN, 64>();
    run<RD32, 64>(); run<RD64, 64>(); run<RD64, 16>(); run<RD128, 64>();
    run<WR64, 64>(); run<WR128, 64>();
    return 0;
}
