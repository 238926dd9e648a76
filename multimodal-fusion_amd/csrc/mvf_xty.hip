// out[a][b] = sum over rows r of x[r][a] * y[r][b]: the weight gradient of a Linear
// (grad_weight = grad_out^T . input) for the dense layers of the attentive pools
// (modules/multimodal/pooling.py:604-685, core/common_modules/base_modules.py:38-48), where
// the reduction runs over millions of rows and the result has at most a few thousand entries.
//
// float32 matrix cores (v_mfma_f32_32x32x2_f32): a wave keeps up to 2 x 2 blocks of 32 x 32
// sums in its accumulators and walks its own contiguous range of rows, two rows per
// instruction.  The A operand of a block is x[row][a0 + lane % 32], the B operand
// y[row][b0 + lane % 32], row = the pair's first row + lane / 32: both are plain dword loads
// of 128 contiguous bytes per row, straight from global memory (one load per operand block
// feeds a 64-cycle instruction, so there is nothing to gain from staging in LDS).  Sixteen
// rows are in flight per wave while it multiplies the sixteen before.  The four waves of a
// workgroup fold their accumulators through LDS in wave order, every workgroup writes one
// partial result and a second kernel adds the partials in workgroup order: no atomics, the
// same bits on every run.  The pass is bound by HBM (4 (A + B) bytes per row).
#include "mvf_common.h"

namespace mvf {

constexpr int XTY_THREADS = 256;
constexpr int XTY_WAVES = XTY_THREADS / 64;
constexpr int XTY_MAX_WGS = 512;
constexpr int XTY_PAIRS = 8;            // row pairs per batch of loads

typedef float xty_acc __attribute__((ext_vector_type(16)));

template <int NA, int NB>
__global__ __launch_bounds__(XTY_THREADS) void xty_kernel(
    const float *__restrict__ x, int ldx, int A, const float *__restrict__ y, int ldy, int B,
    i64 rows, i64 rows_per_wave, int y_ones, float *__restrict__ partial)
{
    __shared__ float fold[NA * NB * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    // columns past the operand's width read column 0: their sums are never stored
    int xo[NA], yo[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) xo[i] = 32 * i + col < A ? 32 * i + col : 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) yo[i] = 32 * i + col < B ? 32 * i + col : 0;
    xty_acc acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const i64 w = (i64)blockIdx.x * XTY_WAVES + wave;
    i64 r = w * rows_per_wave;
    i64 rend = r + rows_per_wave;
    if (rend > rows) rend = rows;
    constexpr int STEP = 2 * XTY_PAIRS;
    float xa[2][XTY_PAIRS][NA], yb[2][XTY_PAIRS][NB];

    auto load_full = [&](int buf, i64 base) {
#pragma unroll
        for (int u = 0; u < XTY_PAIRS; ++u) {
            const i64 row = base + 2 * u + kh;
#pragma unroll
            for (int i = 0; i < NA; ++i) xa[buf][u][i] = x[row * ldx + xo[i]];
#pragma unroll
            for (int i = 0; i < NB; ++i) yb[buf][u][i] = y_ones ? 1.0f : y[row * ldy + yo[i]];
        }
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int u = 0; u < XTY_PAIRS; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[buf][u][i], yb[buf][u][j], acc[i][j], 0, 0, 0);
    };

    if (r + STEP <= rend) {
        load_full(0, r);
        int buf = 0;
        for (; r + 2 * STEP <= rend; r += STEP) {
            if (buf == 0) { load_full(1, r + STEP); multiply(0); }
            else { load_full(0, r + STEP); multiply(1); }
            buf ^= 1;
        }
        if (buf == 0) multiply(0); else multiply(1);
        r += STEP;
    }
    // the last rows of the range, pair by pair; a row past the end counts as zeros
    for (; r < rend; r += 2) {
        const i64 row = r + kh;
        const bool ok = row < rend;
        const i64 rr = ok ? row : r;
        float a[NA], b[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) { a[i] = x[rr * ldx + xo[i]]; a[i] = ok ? a[i] : 0.0f; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { b[i] = y_ones ? 1.0f : y[rr * ldy + yo[i]]; b[i] = ok ? b[i] : 0.0f; }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    // fold of the workgroup's waves, in wave order
    for (int turn = 0; turn < XTY_WAVES; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int at = ((i * NB + j) * 16 + e) * 64 + lane;
                        fold[at] = turn == 0 ? acc[i][j][e] : fold[at] + acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    float *p = partial + (size_t)blockIdx.x * (NA * NB * 16 * 64);
    for (int at = threadIdx.x; at < NA * NB * 16 * 64; at += XTY_THREADS) p[at] = fold[at];
}

// out[a][b] = sum of the workgroups' partial sums.  A block folds 16 consecutive accumulator
// slots: 16 slices of the workgroups per slot (float64 running sums in workgroup order), then
// the slices in order through LDS.
constexpr int XTY_FOLD_SLOTS = 16, XTY_FOLD_SLICES = 16;

__global__ __launch_bounds__(XTY_FOLD_SLOTS * XTY_FOLD_SLICES) void xty_fold_kernel(
    const float *__restrict__ partial, int nwg, int NA, int NB, int A, int B, float *__restrict__ out, int ldo)
{
    __shared__ double red[XTY_FOLD_SLICES][XTY_FOLD_SLOTS];
    const int slot = threadIdx.x % XTY_FOLD_SLOTS, slice = threadIdx.x / XTY_FOLD_SLOTS;
    const int at = blockIdx.x * XTY_FOLD_SLOTS + slot;          // < NA * NB * 1024 (grid is exact)
    const size_t stride = (size_t)NA * NB * 16 * 64;
    const int per = (nwg + XTY_FOLD_SLICES - 1) / XTY_FOLD_SLICES;
    const int k1 = min(nwg, (slice + 1) * per);
    double s = 0.0;
    for (int k = slice * per; k < k1; ++k) s += (double)partial[k * stride + at];
    red[slice][slot] = s;
    __syncthreads();
    if (slice != 0) return;
    for (int k = 1; k < XTY_FOLD_SLICES; ++k) s += red[k][slot];
    // (block, accumulator register e, lane) of the slot -> (a, b)
    const int lane = at & 63, e = (at >> 6) & 15, blk = at >> 10;
    const int a = (blk / NB) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    const int b = (blk % NB) * 32 + (lane & 31);
    if (a < A && b < B) out[(size_t)a * ldo + b] = (float)s;
}

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_xty_workspace_bytes(void)
{
    return align_up((size_t)XTY_MAX_WGS * 4 * 16 * 64 * sizeof(float));
}

extern "C" int mvf_xty(
    const float *x, int32_t ldx, int32_t A, const float *y, int32_t ldy, int32_t B, int64_t rows,
    float *out, int32_t ldo, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(A >= 1 && B >= 1 && rows >= 0, "xty: A and B must be positive, rows non-negative");
    MVF_REQUIRE(out && ldo >= B, "xty: null output or ldo < B");
    MVF_REQUIRE(rows == 0 || (x && ldx >= A && (y ? ldy >= B : B == 1)),
                "xty: null input, a row stride below the width, or B != 1 without y");
    MVF_REQUIRE(workspace && workspace_bytes >= mvf_xty_workspace_bytes(), "xty: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        MVF_HIP(hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)B * sizeof(float), (size_t)A, st));
        return MVF_OK;
    }
    // sixteen rows per batch and wave; at least four batches per wave
    i64 nwg = (rows + XTY_WAVES * 64 - 1) / (XTY_WAVES * 64);
    if (nwg > XTY_MAX_WGS) nwg = XTY_MAX_WGS;
    if (nwg < 1) nwg = 1;
    i64 rpw = (rows + nwg * XTY_WAVES - 1) / (nwg * XTY_WAVES);
    rpw = (rpw + 15) / 16 * 16;
    float *partial = (float *)workspace;
    // tiles of 64 x 64 sums (the operands of a tile are read once per tile)
    for (int a0 = 0; a0 < A; a0 += 64)
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int At = A - a0 < 64 ? A - a0 : 64, Bt = B - b0 < 64 ? B - b0 : 64;
            const int NA = At > 32 ? 2 : 1, NB = Bt > 32 ? 2 : 1;
            const float *xt = x + a0, *yt = y ? y + b0 : x;
#define MVF_XTY_LAUNCH(NA_, NB_)                                                                  \
    hipLaunchKernelGGL((xty_kernel<NA_, NB_>), dim3((unsigned)nwg), dim3(XTY_THREADS), 0, st, xt, \
                       (int)ldx, At, yt, (int)ldy, Bt, (i64)rows, rpw, y ? 0 : 1, partial)
            if (NA == 1 && NB == 1) MVF_XTY_LAUNCH(1, 1);
            else if (NA == 1) MVF_XTY_LAUNCH(1, 2);
            else if (NB == 1) MVF_XTY_LAUNCH(2, 1);
            else MVF_XTY_LAUNCH(2, 2);
#undef MVF_XTY_LAUNCH
            hipLaunchKernelGGL(xty_fold_kernel, dim3(NA * NB * 1024 / XTY_FOLD_SLOTS), dim3(XTY_FOLD_SLOTS * XTY_FOLD_SLICES), 0, st,
                               (const float *)partial, (int)nwg, NA, NB, At, Bt,
                               out + (size_t)a0 * ldo + b0, (int)ldo);
        }
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
