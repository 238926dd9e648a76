// Does LDS atomic throughput scale with resident waves?  (latency-bound per wave?)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
template <int LDS_SLOTS, int INFLIGHT>
__global__ __launch_bounds__(256) void k(int iters, u64 *out)
{
    __shared__ u64 zb[LDS_SLOTS];
    for (int p = threadIdx.x; p < LDS_SLOTS; p += 256) zb[p] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u64 *p = zb + lane;
    u64 key = ((u64)(1000 + blockIdx.x) << 32) | threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) atomicMin(p + u * 64, key);
        if (INFLIGHT == 0) __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) every 8
        key -= 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = zb[5];
}
template <int LDS_SLOTS, int INFLIGHT> void run(int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 2000;
    u64 *out; (void)hipMalloc(&out, blocks * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<LDS_SLOTS, INFLIGHT><<<blocks, 256>>>(iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<LDS_SLOTS, INFLIGHT><<<blocks, 256>>>(iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 8;
    printf("LDS %5d B/WG  wg/cu %d  waitcnt-each-8=%d  %8.3f ms  %.2f cycles/wave-instr/CU @2.4GHz\n",
           LDS_SLOTS * 8, wg_per_cu, INFLIGHT == 0, ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
    (void)hipFree(out);
}
int main()
{
    run<4096, 1>(1); run<4096, 1>(2); run<4096, 1>(5);
    run<2048, 1>(8); run<1024, 1>(8);
    run<4096, 0>(5);
    return 0;
}
