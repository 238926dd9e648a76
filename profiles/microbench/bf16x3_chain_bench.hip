// Two Linear(64 -> 64) + LeakyReLU layers on 32-row tiles with float32 accuracy on the bf16
// matrix cores: every float32 operand is split exactly into three bf16 parts (x = x1 + x2 + x3,
// round-to-nearest at each step) and a product is the sum of the six part products whose weight
// is at least 2^-16 of the whole (1*1, 1*2, 2*1, 2*2, 1*3, 3*1); the dropped ones are below
// 2^-24.  Activations stay in registers between the layers (an accumulator block is the next
// layer's B operand), weights sit in LDS already split and in the lane order of the A operand.
// Prints the time for 2 M rows and the largest error against a float64 reference next to the
// error of a plain float32 evaluation of the same layers.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off bf16x3_chain_bench.hip -o bf16x3_chain_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi)
{
    floatx2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split8(const float *v, uintx4 &p1, uintx4 &p2, uintx4 &p3)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned d1 = cvt_pk(a, b);
        const float ra = __fsub_rn(a, __uint_as_float(d1 << 16)), rb = __fsub_rn(b, __uint_as_float(d1 & 0xffff0000u));
        const unsigned d2 = cvt_pk(ra, rb);
        const float sa = __fsub_rn(ra, __uint_as_float(d2 << 16)), sb = __fsub_rn(rb, __uint_as_float(d2 & 0xffff0000u));
        p1[i] = d1; p2[i] = d2; p3[i] = cvt_pk(sa, sb);
    }
}

__device__ __forceinline__ floatx16 mfma6(const uintx4 &a1, const uintx4 &a2, const uintx4 &a3,
                                          const uintx4 &b1, const uintx4 &b2, const uintx4 &b3, floatx16 acc)
{
#ifdef NO_MFMA
#define MF(A_, B_) acc[0] += __uint_as_float(A_[0] ^ B_[1]); acc[5] += __uint_as_float(A_[2] ^ B_[3])
#else
#define MF(A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), acc, 0, 0, 0)
#endif
    MF(a1, b3); MF(a3, b1); MF(a2, b2); MF(a1, b2); MF(a2, b1); MF(a1, b1);
#undef MF
    return acc;
}

// LDS image of one layer: [kstep 4][out block 2][part 3][lane 64] x 16 bytes
#ifndef WPE
#define WPE 2
#endif
constexpr int LAYER_VEC = 4 * 2 * 3 * 64;

// input channel of element j of lane half kh in k-step q: the first layer reads the lane's
// contiguous half of its row, later layers take accumulator registers 8 (q & 1) .. + 7 of block q >> 1
__host__ __device__ inline int k_first(int q, int kh, int j) { return kh * 32 + 8 * q + j; }
__host__ __device__ inline int k_next(int q, int kh, int j)
{
    const int r = 8 * (q & 1) + j;
    return (q >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
}

#define SB() __builtin_amdgcn_sched_barrier(0)

// The split of 8 values (with the activation of the layer that produced them) as a list of
// single vector instructions in stage-major order: neighbours in the list do not depend on each
// other, so a run of them issues without waiting.  44 operations, 60 with the activation.
struct Split8 {
    float x[8], t[8];
    unsigned d[4], h[4], g[4];
    uintx4 p1, p2, p3;
    __device__ __forceinline__ void op(int i, const float *v, bool act, float slope)
    {
        if (act) {
            if (i < 8) { t[i] = __fmul_rn(v[i], slope); return; }
            if (i < 16) { x[i - 8] = fmaxf(v[i - 8], t[i - 8]); return; }
            i -= 16;
        }
        const int st = i >> 2, e = i & 3;
        const float *src = (st == 0 && !act) ? v : x;
        switch (st) {
        case 0: d[e] = cvt_pk(src[2 * e], src[2 * e + 1]); p1[e] = d[e]; if (!act) { x[2 * e] = v[2 * e]; x[2 * e + 1] = v[2 * e + 1]; } break;
        case 1: case 6: h[e] = d[e] << 16; break;
        case 2: case 7: g[e] = d[e] & 0xffff0000u; break;
        case 3: case 8: x[2 * e] = __fsub_rn(x[2 * e], __uint_as_float(h[e])); break;
        case 4: case 9: x[2 * e + 1] = __fsub_rn(x[2 * e + 1], __uint_as_float(g[e])); break;
        case 5: d[e] = cvt_pk(x[2 * e], x[2 * e + 1]); p2[e] = d[e]; break;
        default: p3[e] = cvt_pk(x[2 * e], x[2 * e + 1]); break;
        }
    }
    // piece i of n
    __device__ __forceinline__ void piece(int i, int n, const float *v, bool act, float slope)
    {
#ifdef NO_FILL
        if (i >= 0) { p1[0] ^= 1; return; }
#endif
        const int total = act ? 60 : 44, per = (total + n - 1) / n;
#pragma unroll
        for (int k = i * per; k < (i + 1) * per; ++k)
            if (k < total) op(k, v, act, slope);
    }
};

__device__ __forceinline__ floatx16 mf(const uintx4 &a, const uintx4 &b, floatx16 acc)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// One k-step into both output blocks: 12 MFMAs, after each of them one piece of `fill`.  The
// weights of this step are in `a`; those of the next step (at `nxt`) are requested first and
// replace them at the end.
template <class F>
__device__ __forceinline__ void kstep2(uintx4 (&a)[6], const uintx4 *nxt, const uintx4 (&b)[3], floatx16 &n0, floatx16 &n1, F fill)
{
    uintx4 t[6];
    SB();
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] = nxt[i * 64];
    SB();
#ifdef ALT
    n0 = mf(a[0], b[2], n0); SB(); fill(0); SB();
    n1 = mf(a[3], b[2], n1); SB(); fill(1); SB();
    n0 = mf(a[2], b[0], n0); SB(); fill(2); SB();
    n1 = mf(a[5], b[0], n1); SB(); fill(3); SB();
    n0 = mf(a[1], b[1], n0); SB(); fill(4); SB();
    n1 = mf(a[4], b[1], n1); SB(); fill(5); SB();
    n0 = mf(a[0], b[1], n0); SB(); fill(6); SB();
    n1 = mf(a[3], b[1], n1); SB(); fill(7); SB();
    n0 = mf(a[1], b[0], n0); SB(); fill(8); SB();
    n1 = mf(a[4], b[0], n1); SB(); fill(9); SB();
    n0 = mf(a[0], b[0], n0); SB(); fill(10); SB();
    n1 = mf(a[3], b[0], n1); SB(); fill(11); SB();
#else
    n0 = mf(a[0], b[2], n0); SB(); fill(0); SB();
    n0 = mf(a[2], b[0], n0); SB(); fill(1); SB();
    n0 = mf(a[1], b[1], n0); SB(); fill(2); SB();
    n0 = mf(a[0], b[1], n0); SB(); fill(3); SB();
    n0 = mf(a[1], b[0], n0); SB(); fill(4); SB();
    n0 = mf(a[0], b[0], n0); SB(); fill(5); SB();
    n1 = mf(a[3], b[2], n1); SB(); fill(6); SB();
    n1 = mf(a[5], b[0], n1); SB(); fill(7); SB();
    n1 = mf(a[4], b[1], n1); SB(); fill(8); SB();
    n1 = mf(a[3], b[1], n1); SB(); fill(9); SB();
    n1 = mf(a[4], b[0], n1); SB(); fill(10); SB();
    n1 = mf(a[3], b[0], n1); SB(); fill(11); SB();
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = t[i];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void chain_kernel(const float *x, const float *w0, const float *w1, float *out, long rows, float slope)
{
    extern __shared__ uintx4 lds[];
    for (int idx = threadIdx.x; idx < 2 * 4 * 2 * 64; idx += 256) {
        const int lane = idx & 63, nbo = (idx >> 6) & 1, q = (idx >> 7) & 3, l = idx >> 9;
        const int m = nbo * 32 + (lane & 31), kh = lane >> 5;
        float v[8];
        for (int j = 0; j < 8; ++j) {
            const int k = l == 0 ? k_first(q, kh, j) : k_next(q, kh, j);
            v[j] = (l == 0 ? w0 : w1)[m * 64 + k];
        }
        uintx4 p1, p2, p3;
        split8(v, p1, p2, p3);
        uintx4 *dst = lds + l * LAYER_VEC + ((q * 2 + nbo) * 3) * 64 + lane;
        dst[0] = p1; dst[64] = p2; dst[128] = p3;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jrow = lane & 31, kh = lane >> 5;
    const long stride = (long)gridDim.x * 128;
    long base = (long)blockIdx.x * 128;
    if (base >= rows) return;
    const long long c0 = clock64(), w0c = wall_clock64();
    int ntiles = 0;
    float xr[32];
    uintx4 cur[3];             // the split k-step about to be multiplied
    {
        const long row = base + wave * 32 + jrow;
        const float *p = x + (row < rows ? row : 0) * 64 + kh * 32;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t = *(const float4 *)(p + 4 * q);
            xr[4 * q] = t.x; xr[4 * q + 1] = t.y; xr[4 * q + 2] = t.z; xr[4 * q + 3] = t.w;
        }
        split8(xr, cur[0], cur[1], cur[2]);
    }
    for (; base < rows; base += stride) {
        ++ntiles;
        const long row = base + wave * 32 + jrow;
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        const uintx4 *a0 = lds + lofs;     // the second layer's image follows the first: k-steps 0 .. 7
        floatx16 acc0 = {0}, acc1 = {0}, n0 = {0}, n1 = {0};
        uintx4 aw[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) aw[i] = a0[i * 64];
        Split8 sp;
        float v8[8];
        // first layer: k-step q's products run beside the split of k-step q + 1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
                kstep2(aw, a0 + ((q + 1) * 6) * 64, cur, acc0, acc1, [&](int i) { sp.piece(i, 12, xr + 8 * (q + 1), false, slope); });
                cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
            } else {
                kstep2(aw, a0 + ((q + 1) * 6) * 64, cur, acc0, acc1, [&](int) {});
            }
        }
        // next tile's rows (rows past the end re-read row 0)
        {
            const long nrow = row + stride;
            const float *p = x + (nrow < rows ? nrow : 0) * 64 + kh * 32;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#ifdef NO_LOAD
                const float4 t = make_float4(xr[4 * q + 1], xr[4 * q], xr[4 * q + 3], xr[4 * q + 2]); (void)p;
#else
                const float4 t = *(const float4 *)(p + 4 * q);
#endif
                xr[4 * q] = t.x; xr[4 * q + 1] = t.y; xr[4 * q + 2] = t.z; xr[4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc0[j];
#pragma unroll
        for (int i = 0; i < 12; ++i) sp.piece(i, 12, v8, true, slope);
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = ((q + 1) >> 1) ? acc1[8 * ((q + 1) & 1) + j] : acc0[8 * ((q + 1) & 1) + j];
                kstep2(aw, a0 + ((q + 5) * 6) * 64, cur, n0, n1, [&](int i) { sp.piece(i, 12, v8, true, slope); });
            } else {
                kstep2(aw, a0, cur, n0, n1, [&](int i) { sp.piece(i, 12, xr, false, slope); });
            }
            cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        }
#ifdef NO_STORE
        if (n0[0] == 1234.5f && n1[3] == 77.f) {
#else
        {
#endif
            float *o = out + row * 64 + 4 * kh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *(float4 *)(o + 8 * g) = make_float4(n0[4 * g], n0[4 * g + 1], n0[4 * g + 2], n0[4 * g + 3]);
                *(float4 *)(o + 32 + 8 * g) = make_float4(n1[4 * g], n1[4 * g + 1], n1[4 * g + 2], n1[4 * g + 3]);
            }
        }
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        const long long c1 = clock64(), w1c = wall_clock64();
        printf("block 7 wave 0: %d tiles, %lld cycles per tile, %.0f MHz (100 MHz wall clock)\n", ntiles, (c1 - c0) / ntiles, (double)(c1 - c0) / (double)(w1c - w0c) * 100.0);
    }
}

int main()
{
    const long rows = 2000000 / 128 * 128;
    std::vector<float> x(rows * 64), w0(64 * 64), w1(64 * 64);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : x) v = rnd() * 3.f;
    for (auto &v : w0) v = rnd() * 0.3f;
    for (auto &v : w1) v = rnd() * 0.3f;
    float *dx, *dw0, *dw1, *dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw0, 64 * 64 * 4); hipMalloc(&dw1, 64 * 64 * 4); hipMalloc(&dout, x.size() * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw0, w0.data(), 64 * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw1, w1.data(), 64 * 64 * 4, hipMemcpyHostToDevice);
    const size_t lds = 2 * LAYER_VEC * 16;
    hipFuncSetAttribute((const void *)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int per_cu : {1, 2, 3}) {
        const int grid = 256 * per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        chain_kernel<<<grid, 256, lds>>>(dx, dw0, dw1, dout, rows, 0.2f);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) chain_kernel<<<grid, 256, lds>>>(dx, dw0, dw1, dout, rows, 0.2f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        printf("workgroups/CU %d: %.1f us per launch, %.1f TFLOP/s float32-equivalent, %.2f TB/s\n", per_cu, ms * 1e3,
               rows * 2.0 * 2 * 64 * 64 / ms / 1e9, rows * 512.0 / ms / 1e9);
    }
    std::vector<float> out(4096 * 64);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0, mag = 0;
    for (int r = 0; r < 4096; ++r) {
        double h[64]; float hf[64];
        for (int m = 0; m < 64; ++m) {
            double s = 0; float sf = 0.f;
            for (int k = 0; k < 64; ++k) { s += (double)x[r * 64 + k] * w0[m * 64 + k]; sf = fmaf(x[r * 64 + k], w0[m * 64 + k], sf); }
            // the kernel's second layer starts from the float32 result of the first
            h[m] = s > 0 ? s : 0.2 * s; hf[m] = sf > 0 ? sf : 0.2f * sf;
        }
        for (int m = 0; m < 64; ++m) {
            double s = 0; float sf = 0.f;
            for (int k = 0; k < 64; ++k) { s += h[k] * w1[m * 64 + k]; sf = fmaf(hf[k], w1[m * 64 + k], sf); }
            worst = fmax(worst, fabs(out[r * 64 + m] - s));
            worst32 = fmax(worst32, fabs((double)sf - s));
            mag = fmax(mag, fabs(s));
        }
    }
    printf("largest |error| vs float64: bf16x3 %.3g, plain float32 %.3g (largest magnitude %.3g)\n", worst, worst32, mag);
    return 0;
}
