// y = x W^T (+ b) over many contiguous float32 rows with at most 64 input and 64 output
// columns: the output and the input gradient of a torch.nn.Linear in the differentiable path of
// the attentive pools (core/common_modules/base_modules.py:38-48, modules/multimodal/
// pooling.py:406-452), where neither BatchNorm nor an activation can be folded into the launch
// (batch statistics come between them).  A pure streaming pass: 4 (I + O) bytes per row.
//
// float32 matrix cores (v_mfma_f32_32x32x2_f32), one workgroup of four waves per compute unit
// (one wave per SIMD), tiles of 32 rows dealt to the waves in turn.  A tile of x is one
// contiguous piece of memory: the wave reads it with 16-byte loads in memory order (1 KB per
// instruction), parks it in its own LDS region with a row pitch of 16 bytes more than the row,
// and reads it back as the B operand (lane = row, each half of the wave a contiguous half of
// the row: 16-byte LDS reads without bank conflicts).  The weights sit in registers as the A
// operand for the whole launch (lane = output channel, k = the lane half's column).  The 32 x O
// results go the same way back: accumulators to LDS as rows, LDS to memory in memory order.
// The next tile's loads are in flight while the current one is multiplied.
#include "mvf_common.h"

namespace mvf {

constexpr int SL_THREADS = 256;
constexpr int SL_WAVES = SL_THREADS / 64;
constexpr int SL_ROWS = 32;                 // rows per tile
constexpr int SL_MAX_C = 64;
constexpr int SL_PITCH_MAX = SL_MAX_C * 4 + 16;
constexpr int SL_MAX_WGS = 256;             // workgroups of a launch (one per compute unit), 4 partial rows each

typedef float sl_acc __attribute__((ext_vector_type(16)));
typedef float sl_f4 __attribute__((ext_vector_type(4)));

// KQ: 16-byte pieces of a lane half's part of an input row (I = 8 KQ); NO: blocks of 32 output
// channels.
// STATS: the column sums of y and y * y over the wave's rows as well (float32 per lane over its
// tiles, float64 across the lanes), one partial row [2][O] per wave: what BatchNorm's batch
// statistics are folded from (mvf_batch_norm_act_forward's col_sums).
template <int KQ, int NO, bool STATS>
__global__ __launch_bounds__(SL_THREADS) void stream_linear_kernel(
    const float *__restrict__ x, int I, const float *__restrict__ W, const float *__restrict__ bias, int O,
    i64 rows, float *__restrict__ y, double *__restrict__ col_sums)
{
    // one region per wave: a tile of x on its way in, then the tile of y on its way out (the
    // LDS operations of a wave execute in order)
    __shared__ __attribute__((aligned(16))) char lds[SL_WAVES][SL_ROWS * SL_PITCH_MAX];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int half = I >> 1;                        // columns of a lane half
    const int iq = I >> 2, oq = O >> 2;             // 16-byte pieces of an input / output row
    const unsigned pitch_i = (unsigned)I * 4u + 16u, pitch_o = (unsigned)O * 4u + 16u;
    char *lin = lds[wave], *lout = lds[wave];

    // weights: lane (m = n, kh) holds W[32 o + m][kh * half + s] for step s
    float w[NO][4 * KQ];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int s = 0; s < 4 * KQ; ++s) {
            const int ch = 32 * o + n;
            w[o][s] = ch < O ? W[(size_t)ch * I + kh * half + s] : 0.0f;
        }
    // bias of the lane's accumulator registers
    float bz[NO][16];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ch = 32 * o + (e & 3) + 8 * (e >> 2) + 4 * kh;
            bz[o][e] = (bias && ch < O) ? bias[ch] : 0.0f;
        }

    // memory order <-> LDS rows: piece f = lane + 64 j of a tile is row f / iq, piece f % iq
    unsigned in_at[KQ], out_at[4 * NO];
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const unsigned f = (unsigned)lane + 64u * j, r = f / (unsigned)iq;
        in_at[j] = r * pitch_i + (f - r * iq) * 16u;
    }
#pragma unroll
    for (int j = 0; j < 4 * NO; ++j) {
        const unsigned f = (unsigned)lane + 64u * j, r = f / (unsigned)oq;
        out_at[j] = r * pitch_o + (f - r * oq) * 16u;
    }

    const i64 ntiles = (rows + SL_ROWS - 1) / SL_ROWS;
    const i64 wid = (i64)blockIdx.x * SL_WAVES + wave, nwaves = (i64)gridDim.x * SL_WAVES;
    sl_f4 g[KQ];

    // 16-byte pieces of tile t in memory order (a piece past the last row reads the last piece
    // there is: its row's results are never stored)
    auto load_tile = [&](i64 t) {
        const int last = (int)(rows - t * SL_ROWS < SL_ROWS ? rows - t * SL_ROWS : SL_ROWS) * iq - 1;
        const sl_f4 *src = (const sl_f4 *)(x + (size_t)t * SL_ROWS * I);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int f = lane + 64 * j;
            g[j] = src[f < last ? f : last];
        }
    };

    // float64 from the first addition on: sum(y^2) / n - mean^2 cancels when a column's mean is
    // large against its spread, and float32 partial sums would not survive that
    double sum1[STATS ? NO : 1][16], sum2[STATS ? NO : 1][16];
    if constexpr (STATS) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum1[o][e] = sum2[o][e] = 0.0;
    }

    i64 t = wid;
    if (t < ntiles) load_tile(t);
    for (; t < ntiles; t += nwaves) {
        // the tile into LDS as rows, then the lane's half row out of it
#pragma unroll
        for (int j = 0; j < KQ; ++j) *(sl_f4 *)(lin + in_at[j]) = g[j];
        if (t + nwaves < ntiles) load_tile(t + nwaves);
        float xv[4 * KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const sl_f4 v = *(const sl_f4 *)(lin + (unsigned)n * pitch_i + (unsigned)(kh * half) * 4u + 16u * q);
            xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
        }
        sl_acc acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[o][e] = bz[o][e];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4 * KQ; ++s)
#pragma unroll
            for (int o = 0; o < NO; ++o)
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[o][s], xv[s], acc[o], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STATS) {
            // (the rows past the end of the last tile hold copies of its last piece)
            const float live = (i64)n < rows - t * SL_ROWS ? 1.0f : 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const double v = (double)(acc[o][e] * live);
                    sum1[o][e] += v;
                    sum2[o][e] += v * v;
                }
        }
        // results to LDS as rows (the lane's row n, 4 adjacent channels per store), then to memory
        // in memory order
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * o + 8 * q + 4 * kh;
                if (ch < O) {
                    const sl_f4 v = {acc[o][4 * q], acc[o][4 * q + 1], acc[o][4 * q + 2], acc[o][4 * q + 3]};
                    *(sl_f4 *)(lout + (unsigned)n * pitch_o + (unsigned)ch * 4u) = v;
                }
            }
        const int opieces = (int)(rows - t * SL_ROWS < SL_ROWS ? rows - t * SL_ROWS : SL_ROWS) * oq;
        sl_f4 *dst = (sl_f4 *)(y + (size_t)t * SL_ROWS * O);
        if (opieces == 64 * 4 * NO) {       // a whole tile of 32 NO output channels: no lane is idle
#pragma unroll
            for (int j = 0; j < 4 * NO; ++j) dst[lane + 64 * j] = *(const sl_f4 *)(lout + out_at[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4 * NO; ++j) {
                const int f = lane + 64 * j;
                if (f < opieces) dst[f] = *(const sl_f4 *)(lout + out_at[j]);
            }
        }
    }
    if constexpr (STATS) {
        // sums over the 32 rows (lanes) of each half of the wave, in float64, a fixed tree
        double *out = col_sums + (size_t)wid * 2 * O;
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                double a = sum1[o][e], b = sum2[o][e];
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    a += __shfl_xor(a, d, 64);
                    b += __shfl_xor(b, d, 64);
                }
                const int ch = 32 * o + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (n == 0 && ch < O) {
                    out[ch] = a;
                    out[O + ch] = b;
                }
            }
    }
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_linear_rows_supported(int32_t I, int32_t O)
{
    return I >= 8 && I <= SL_MAX_C && I % 8 == 0 && O >= 4 && O <= SL_MAX_C && O % 4 == 0;
}

static int linear_rows_launch(
    const float *x, int32_t I, const float *weight, const float *bias, int32_t O, int64_t rows,
    float *y, double *col_sums, int32_t *n_col_sums, void *stream)
{
    MVF_REQUIRE(mvf_linear_rows_supported(I, O),
                "linear_rows: I must be a multiple of 8 and O a multiple of 4, both at most 64 (got %d -> %d)", I, O);
    MVF_REQUIRE(rows >= 0, "linear_rows: negative row count");
    if (n_col_sums) *n_col_sums = 0;
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(x && weight && y, "linear_rows: null argument");
    MVF_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "linear_rows: x and y must be 16-byte aligned");
    static int cu_cache[64];
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_cache[dev] == 0) {
            int n = 0;
            cu_cache[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
        }
        cus = cu_cache[dev];
    }
    const i64 ntiles = (rows + SL_ROWS - 1) / SL_ROWS;
    i64 nwg = (ntiles + SL_WAVES - 1) / SL_WAVES;
    if (nwg > cus) nwg = cus;
    if (nwg > SL_MAX_WGS) nwg = SL_MAX_WGS;
    if (n_col_sums) *n_col_sums = (int32_t)(nwg * SL_WAVES);
    hipStream_t st = (hipStream_t)stream;
    const int kq = I / 8, no = O > 32 ? 2 : 1;
#define MVF_SL_LAUNCH(KQ_, NO_, ST_)                                                               \
    hipLaunchKernelGGL((stream_linear_kernel<KQ_, NO_, ST_>), dim3((unsigned)nwg), dim3(SL_THREADS), 0, st, \
                       x, (int)I, weight, bias, (int)O, (i64)rows, y, col_sums)
#define MVF_SL_LAUNCH_NO(KQ_)                                                                      \
    case KQ_:                                                                                      \
        if (col_sums) { if (no == 1) MVF_SL_LAUNCH(KQ_, 1, true); else MVF_SL_LAUNCH(KQ_, 2, true); } \
        else { if (no == 1) MVF_SL_LAUNCH(KQ_, 1, false); else MVF_SL_LAUNCH(KQ_, 2, false); }      \
        break
    switch (kq) {
    MVF_SL_LAUNCH_NO(1); MVF_SL_LAUNCH_NO(2); MVF_SL_LAUNCH_NO(3); MVF_SL_LAUNCH_NO(4);
    MVF_SL_LAUNCH_NO(5); MVF_SL_LAUNCH_NO(6); MVF_SL_LAUNCH_NO(7); default: MVF_SL_LAUNCH_NO(8);
    }
#undef MVF_SL_LAUNCH_NO
#undef MVF_SL_LAUNCH
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_linear_rows(
    const float *x, int32_t I, const float *weight, const float *bias, int32_t O, int64_t rows,
    float *y, void *stream)
{
    return linear_rows_launch(x, I, weight, bias, O, rows, y, nullptr, nullptr, stream);
}

extern "C" size_t mvf_linear_rows_sums_bytes(int32_t O)
{
    return O < 1 ? 0 : align_up((size_t)SL_MAX_WGS * SL_WAVES * 2 * (size_t)O * sizeof(double));
}

extern "C" int mvf_linear_rows_sums(
    const float *x, int32_t I, const float *weight, const float *bias, int32_t O, int64_t rows,
    float *y, double *col_sums, size_t col_sums_bytes, int32_t *n_col_sums, void *stream)
{
    MVF_REQUIRE(col_sums && n_col_sums && col_sums_bytes >= mvf_linear_rows_sums_bytes(O),
                "linear_rows_sums: null or too small buffer for the column sums");
    return linear_rows_launch(x, I, weight, bias, O, rows, y, col_sums, n_col_sums, stream);
}
