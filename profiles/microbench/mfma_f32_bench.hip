// Throughput of v_mfma_f32_32x32x2_f32 streams: NACC independent accumulators,
// operands from registers or (LDS = 1) one ds_read_b32 per MFMA, 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_f32_bench.hip -o mfma_f32_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float w[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 256) w[i] = 0.001f * (i & 63);
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 0.01f;
    const float *wp = w + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const float b = LDS ? wp[((s * NACC + a) & 63) * 64] : x;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, x, acc[a], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int LDS> void run(int wgs_per_cu, float *out)
{
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS><<<grid, 256>>>(out, 10);
    hipEventRecord(e0);
    k<NACC, LDS><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("acc %d lds %d waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s\n", NACC, LDS, wgs_per_cu, ms, flops / ms / 1e9);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<1, 0>(w, out); run<2, 0>(w, out); run<4, 0>(w, out);
        run<1, 1>(w, out); run<2, 1>(w, out); run<4, 1>(w, out);
    }
    return 0;
}
