// How many vector instructions hide in the shadow of one v_mfma_f32_32x32x16_bf16?
// One wave per SIMD runs a dependent MFMA chain with N independent vector instructions of one
// kind written after every MFMA; prints cycles per MFMA for N = 0 .. 8.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters)
{
    floatx16 acc = {0};
    uintx4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    float f[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            }
        }
    }
    const long long c1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += f[i] + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 3) *cyc = (c1 - c0) / (iters * 8);
}

template <int N, int KIND> void run(float *out, long long *cyc, int wgs)
{
    k<N, KIND><<<256 * wgs, 256>>>(out, cyc, 2000);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf(" %lld", c);
}

template <int KIND> void row(const char *name, float *out, long long *cyc, int wgs)
{
    printf("%-22s waves/SIMD %d:", name, wgs);
    run<0, KIND>(out, cyc, wgs); run<1, KIND>(out, cyc, wgs); run<2, KIND>(out, cyc, wgs); run<3, KIND>(out, cyc, wgs);
    run<4, KIND>(out, cyc, wgs); run<5, KIND>(out, cyc, wgs); run<6, KIND>(out, cyc, wgs); run<7, KIND>(out, cyc, wgs);
    run<8, KIND>(out, cyc, wgs);
    printf("   cycles per MFMA with 0 .. 8 fillers\n");
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w : {1, 2}) {
        row<0>("v_add_f32", out, cyc, w);
        row<1>("v_cvt_pk_bf16_f32", out, cyc, w);
        row<2>("v_and_b32 (literal)", out, cyc, w);
        row<3>("v_lshlrev_b32", out, cyc, w);
    }
    return 0;
}
