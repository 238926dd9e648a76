// out[a][b] = sum over rows r of x[r][a] * y[r][b]: the weight gradient of a Linear
// (grad_weight = grad_out^T . input) for the dense layers of the attentive pools
// (modules/multimodal/pooling.py:604-685, core/common_modules/base_modules.py:38-48), where
// the reduction runs over millions of rows and the result has at most a few thousand entries.
//
// float32 matrix cores (v_mfma_f32_32x32x2_f32): a wave keeps up to 2 x 2 blocks of 32 x 32
// sums in its accumulators and takes two rows per instruction.  The A operand of a block is
// x[row][a0 + lane % 32], the B operand y[row][b0 + lane % 32], row = the pair's first row +
// lane / 32: the operands come straight from global memory in the layout the instruction wants,
// 128 contiguous bytes per row and block (one load per operand block feeds a 64-cycle
// instruction, so there is nothing to gain from staging in LDS).  Batches of 16 or 32 rows are
// dealt to the waves in turn; a wave has one batch in flight while it multiplies the batch
// before.  The four waves of a workgroup fold their accumulators through LDS in wave order,
// every workgroup writes one partial result and a second kernel adds the partials in workgroup
// order: no atomics, the same bits on every run.  The pass is bound by HBM (4 (A + B) bytes per
// row).
#include "mvf_common.h"

namespace mvf {

constexpr int XTY_THREADS = 256;
constexpr int XTY_WAVES = XTY_THREADS / 64;
constexpr int XTY_MAX_WGS = 1024;       // partial results in the workspace

typedef float xty_acc __attribute__((ext_vector_type(16)));
typedef float xty_f2 __attribute__((ext_vector_type(2)));
// pointers into global memory that went through integer arithmetic (a generic pointer would load
// with flat instructions, which wait on both memory counters)
typedef const __attribute__((address_space(1))) char *XtyGlobal;
typedef const __attribute__((address_space(1))) float *XtyGlobalF;

// Row pairs in flight per wave and batch (2 x 2 blocks hold 64 accumulators and 4 operand
// registers per row pair), and workgroups per compute unit.
// One workgroup per compute unit, i.e. one wave per SIMD: measured on 2 M rows
// (profiles/microbench/xty_bench.hip), 64 x 64 sums take 150 us with one wave per SIMD and 260-290
// with two or three, whatever the batch (4 to 16 pairs); 32 x 32: 70 against 105; the memory side
// alone (no MFMAs) runs at 6.5 TB/s and the matrix side alone at 115 TFLOP/s with two.
#ifndef MVF_XTY_PAIRS_4
#define MVF_XTY_PAIRS_4 8       // 2 x 2 blocks
#define MVF_XTY_PAIRS_2 8       // 1 x 2, 2 x 1
#define MVF_XTY_PAIRS_1 16      // 1 x 1
#define MVF_XTY_WGS_4 1         // workgroups per compute unit, 2 x 2 blocks
#define MVF_XTY_WGS_1 1         // workgroups per compute unit otherwise
#endif
template <int NA, int NB> struct XtyPairs {
    static constexpr int value = NA * NB == 4 ? MVF_XTY_PAIRS_4 : NA * NB == 2 ? MVF_XTY_PAIRS_2 : MVF_XTY_PAIRS_1;
};

// An operand of two blocks (33 to 64 columns) whose rows are 8-byte aligned is read with one
// 8-byte load per lane: the lane's two blocks are then columns 2 m and 2 m + 1 instead of m and
// 32 + m (VX / VY; the fold kernel undoes the permutation).
template <int N, bool V> struct XtyOperand {
    unsigned off[V ? 1 : N];        // lane offsets in elements (the lane's row of the pair + column)
    __device__ __forceinline__ void init(int kh, int col, int ld, int width)
    {
        if constexpr (V) {
            off[0] = (unsigned)(kh * ld + (2 * col + 1 < width ? 2 * col : 0));
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) off[i] = (unsigned)(kh * ld + (32 * i + col < width ? 32 * i + col : 0));
        }
    }
    __device__ __forceinline__ void load(XtyGlobal row, float (&v)[N], unsigned back = 0u) const
    {
        if constexpr (V) {
            const xty_f2 t = *(const __attribute__((address_space(1))) xty_f2 *)(row + 4u * (off[0] - back));
            v[0] = t.x;
            v[1] = t.y;
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = *(XtyGlobalF)(row + 4u * (off[i] - back));
        }
    }
};

template <int NA, int NB, bool ONES, bool VX, bool VY>
__global__ __launch_bounds__(XTY_THREADS) void xty_kernel(
    const float *__restrict__ x, int ldx, int A, const float *__restrict__ y, int ldy, int B,
    i64 rows, float *__restrict__ partial)
{
    static_assert((!VX || NA == 2) && (!VY || NB == 2) && !(ONES && (VY || NB != 1)), "operand forms");
    __shared__ float fold[NA * NB * 16 * 64];
    // (the wave index as a scalar: row numbers and row addresses are then scalar too, and a load
    // takes its lane offset from one register per operand)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, kh = lane >> 5;
    // columns past the operand's width read column 0: their sums are never stored
    XtyOperand<NA, VX> xop;
    XtyOperand<NB, VY> yop;
    xop.init(kh, col, ldx, A);
    yop.init(kh, col, ldy, B);
    xty_acc acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    constexpr int XTY_PAIRS = XtyPairs<NA, NB>::value;
    constexpr int STEP = 2 * XTY_PAIRS;
    // Batches of STEP rows are dealt to the waves of the launch in turn: at any moment the waves
    // read one contiguous window of the operands, spread over every memory channel (a contiguous
    // range of rows per wave puts the waves a fixed distance apart, on few channels at a time).
    const i64 w = (i64)blockIdx.x * XTY_WAVES + wave, nwaves = (i64)gridDim.x * XTY_WAVES;
    const i64 nbatch = rows / STEP;
    typedef float XBatch[XTY_PAIRS][NA];
    typedef float YBatch[XTY_PAIRS][NB];
    XBatch xa0, xa1;
    YBatch yb0, yb1;

    // The address of a pair's first row is scalar (taken through readfirstlane so that it stays in
    // scalar registers instead of being folded into per-load vector offsets); the lane adds its
    // byte offset.
    auto row_ptr = [](const float *base, i64 row, int ld) {
        const uintptr_t p = (uintptr_t)(base + row * ld);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return (XtyGlobal)(((uintptr_t)hi << 32) | lo);
    };
    auto load_full = [&](XBatch &xa, YBatch &yb, i64 base) {
#pragma unroll
        for (int u = 0; u < XTY_PAIRS; ++u) {
#ifdef MVF_XTY_NO_LOAD      // (profiles/microbench/xty_bench.hip: the matrix-core side alone)
#pragma unroll
            for (int i = 0; i < NA; ++i) xa[u][i] = (float)(base + u);
#pragma unroll
            for (int i = 0; i < NB; ++i) yb[u][i] = (float)(base - u);
#else
            xop.load(row_ptr(x, base + 2 * u, ldx), xa[u]);
            if constexpr (ONES) yb[u][0] = 1.0f;
            else yop.load(row_ptr(y, base + 2 * u, ldy), yb[u]);
#endif
        }
    };
    auto multiply = [&](const XBatch &xa, const YBatch &yb) {
#pragma unroll
        for (int u = 0; u < XTY_PAIRS; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#ifdef MVF_XTY_NO_MFMA      // (profiles/microbench/xty_bench.hip: the memory side alone)
                    acc[i][j][0] += xa[u][i] * yb[u][j];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u][i], yb[u][j], acc[i][j], 0, 0, 0);
#endif
    };

    // two batches alternate: the loads of one are in flight while the other is multiplied
    // (fences: left alone, the scheduler sinks the loads to their uses to save registers and
    // the wave then waits for every batch it has just asked for)
#ifdef MVF_XTY_NO_FENCE
#define XTY_FENCE()
#else
#define XTY_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
    i64 b = w;
    if (b < nbatch) {
        load_full(xa0, yb0, b * STEP);
        b += nwaves;
        for (; b + nwaves < nbatch; b += 2 * nwaves) {
            XTY_FENCE();
            load_full(xa1, yb1, b * STEP);
            XTY_FENCE();
            multiply(xa0, yb0);
            XTY_FENCE();
            load_full(xa0, yb0, (b + nwaves) * STEP);
            XTY_FENCE();
            multiply(xa1, yb1);
            XTY_FENCE();
        }
        if (b < nbatch) {
            load_full(xa1, yb1, b * STEP);
            XTY_FENCE();
            multiply(xa0, yb0);
            multiply(xa1, yb1);
        } else {
            multiply(xa0, yb0);
        }
    }
#undef XTY_FENCE
    // the rows behind the last whole batch, pair by pair, by the wave whose turn it would be; a
    // row past the end counts as zeros (it reads the pair's first row)
    if (w == nbatch % nwaves)
        for (i64 r = nbatch * STEP; r < rows; r += 2) {
            const bool ok = r + kh < rows;
            float a[NA], b2[NB];
            xop.load((XtyGlobal)(x + r * ldx), a, ok ? 0u : (unsigned)(kh * ldx));
            if constexpr (ONES) b2[0] = 1.0f;
            else yop.load((XtyGlobal)(y + r * ldy), b2, ok ? 0u : (unsigned)(kh * ldy));
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = ok ? a[i] : 0.0f;
#pragma unroll
            for (int i = 0; i < NB; ++i) b2[i] = ok ? b2[i] : 0.0f;
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b2[j], acc[i][j], 0, 0, 0);
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
    const float *__restrict__ partial, int nwg, int NA, int NB, int vx, int vy, int A, int B,
    float *__restrict__ out, int ldo)
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
    const int m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), n = lane & 31;
    const int a = vx ? 2 * m + blk / NB : (blk / NB) * 32 + m;
    const int b = vy ? 2 * n + blk % NB : (blk % NB) * 32 + n;
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
    // compute units of the current device (looked up once per device)
    static int cu_cache[64];
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_cache[dev] == 0) {
            int n = 0;
            cu_cache[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
        }
        cus = cu_cache[dev];
    }
    float *partial = (float *)workspace;
    // tiles of 64 x 64 sums (the operands of a tile are read once per tile)
    for (int a0 = 0; a0 < A; a0 += 64)
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int At = A - a0 < 64 ? A - a0 : 64, Bt = B - b0 < 64 ? B - b0 : 64;
            const int NA = At > 32 ? 2 : 1, NB = Bt > 32 ? 2 : 1;
            // one workgroup per compute unit, at least 256 rows each
            i64 nwg = (rows + XTY_WAVES * 64 - 1) / (XTY_WAVES * 64);
            const i64 cap = (i64)cus * (NA * NB == 4 ? MVF_XTY_WGS_4 : MVF_XTY_WGS_1);
            if (nwg > cap) nwg = cap;
            if (nwg > XTY_MAX_WGS) nwg = XTY_MAX_WGS;
            if (nwg < 1) nwg = 1;
            const float *xt = x + a0, *yt = y ? y + b0 : x;
            // 8-byte loads for an operand of two blocks with an even width and 8-byte aligned rows
            const bool vx = NA == 2 && At % 2 == 0 && ldx % 2 == 0 && (uintptr_t)xt % 8 == 0;
            const bool vy = y && NB == 2 && Bt % 2 == 0 && ldy % 2 == 0 && (uintptr_t)yt % 8 == 0;
#define MVF_XTY_LAUNCH(NA_, NB_, ONES_, VX_, VY_)                                                 \
    hipLaunchKernelGGL((xty_kernel<NA_, NB_, ONES_, VX_, VY_>), dim3((unsigned)nwg), dim3(XTY_THREADS), 0, st, xt, \
                       (int)ldx, At, yt, (int)ldy, Bt, (i64)rows, partial)
            if (!y) {
                if (NA == 1) MVF_XTY_LAUNCH(1, 1, true, false, false);
                else if (vx) MVF_XTY_LAUNCH(2, 1, true, true, false);
                else MVF_XTY_LAUNCH(2, 1, true, false, false);
            } else if (NA == 1 && NB == 1) MVF_XTY_LAUNCH(1, 1, false, false, false);
            else if (NA == 1) { if (vy) MVF_XTY_LAUNCH(1, 2, false, false, true); else MVF_XTY_LAUNCH(1, 2, false, false, false); }
            else if (NB == 1) { if (vx) MVF_XTY_LAUNCH(2, 1, false, true, false); else MVF_XTY_LAUNCH(2, 1, false, false, false); }
            else if (vx && vy) MVF_XTY_LAUNCH(2, 2, false, true, true);
            else if (vx) MVF_XTY_LAUNCH(2, 2, false, true, false);
            else if (vy) MVF_XTY_LAUNCH(2, 2, false, false, true);
            else MVF_XTY_LAUNCH(2, 2, false, false, false);
#undef MVF_XTY_LAUNCH
            hipLaunchKernelGGL(xty_fold_kernel, dim3(NA * NB * 1024 / XTY_FOLD_SLOTS), dim3(XTY_FOLD_SLOTS * XTY_FOLD_SLICES), 0, st,
                               (const float *)partial, (int)nwg, NA, NB, (int)vx, (int)vy, At, Bt,
                               out + (size_t)a0 * ldo + b0, (int)ldo);
        }
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
