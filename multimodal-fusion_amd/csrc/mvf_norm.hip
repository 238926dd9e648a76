// Tail of a dense layer in a differentiable forward: BatchNorm1d (batch statistics in
// train mode, running statistics otherwise) + LeakyReLU on float32 rows [rows][C], forward
// and backward (the MLP blocks of core/common_modules/base_modules.py:38-48 with the
// FastBatchNorm1d of :131-156; the Linear in front of them is a plain library GEMM).
//
// Everything here is bound by HBM: the forward reads the rows twice (column sums, then
// normalise + activate) and writes them once; the backward reads rows and gradient twice
// and writes the input gradient once.  Nothing but the input of the layer is kept for
// the backward: the normalised value and the sign of the activation are recomputed.
//
// Layout: with C a multiple of 4 a thread owns one group of 4 adjacent columns for the
// whole launch (its statistics and affine terms stay in registers) and steps down the
// rows; consecutive threads read consecutive 16 bytes.  Column sums are accumulated in
// float64 per thread, reduced per block in a fixed order and written as one partial row
// per block; a second small kernel folds the partials in a fixed order, so the
// statistics are the same on every run.
#include "mvf_common.h"

namespace mvf {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 1024;
constexpr int BN_FOLD_COLS = 4;         // columns per block of the fold kernels

template <int VW> __device__ __forceinline__ void bn_load(const float *p, float (&v)[VW])
{
    if constexpr (VW == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

template <int VW> __device__ __forceinline__ void bn_store(float *p, const float (&v)[VW])
{
    if constexpr (VW == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

// What a thread of the row-walking kernels owns: column group `cv` of its chunk of at most
// BN_THREADS groups (blockIdx.y), rows r0 + k * rstep below `rend`.
struct BnWalk {
    int col;            // first column of the thread's group, -1: idle thread
    i64 r0, rend;
    int rstep;
};

template <int VW> __device__ __forceinline__ BnWalk bn_walk(i64 rows, int C, i64 rows_per_block)
{
    const int CV = C / VW;
    const int chunk0 = blockIdx.y * BN_THREADS;
    const int cw = min(BN_THREADS, CV - chunk0);            // groups in this chunk
    const int rpp = BN_THREADS / cw;                        // rows per pass of the block
    const int t = threadIdx.x;
    BnWalk w;
    w.rstep = rpp;
    const i64 b0 = (i64)blockIdx.x * rows_per_block;
    w.rend = min(rows, b0 + rows_per_block);
    if (t >= rpp * cw) {
        w.col = -1;
        w.r0 = w.rend;
    } else {
        w.col = (chunk0 + t % cw) * VW;
        w.r0 = b0 + t / cw;
    }
    return w;
}

// Per-column terms a thread keeps in registers.
template <int VW> struct BnTerms {
    float mean[VW], invstd[VW], gamma[VW], beta[VW];
    __device__ __forceinline__ void load(int col, const float *save_mean, const float *save_invstd,
                                         const float *weight, const float *bias)
    {
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            mean[j] = save_mean[col + j];
            invstd[j] = save_invstd[col + j];
            gamma[j] = weight ? weight[col + j] : 1.0f;
            beta[j] = bias ? bias[col + j] : 0.0f;
        }
    }
};

// Block-level fold of per-thread float64 pairs into one partial row:
// partial[block][0][C] and partial[block][1][C].
template <int VW>
__device__ __forceinline__ void bn_block_fold(const double (&s)[VW], const double (&q)[VW], int C, double *partial)
{
    __shared__ double red[BN_THREADS][2 * VW];
    const int CV = C / VW;
    const int chunk0 = blockIdx.y * BN_THREADS;
    const int cw = min(BN_THREADS, CV - chunk0);
    const int rpp = BN_THREADS / cw;
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        red[t][j] = s[j];
        red[t][VW + j] = q[j];
    }
    __syncthreads();
    if (t < cw) {
        double a[2 * VW];
#pragma unroll
        for (int j = 0; j < 2 * VW; ++j) a[j] = red[t][j];
        for (int r = 1; r < rpp; ++r)
#pragma unroll
            for (int j = 0; j < 2 * VW; ++j) a[j] += red[r * cw + t][j];
        double *p = partial + (size_t)blockIdx.x * 2 * C;
        const int col = (chunk0 + t) * VW;
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            p[col + j] = a[j];
            p[C + col + j] = a[VW + j];
        }
    }
}

// Column sums of z and z*z.
template <int VW>
__global__ __launch_bounds__(BN_THREADS) void bn_sums_kernel(
    const float *__restrict__ z, i64 rows, int C, i64 rows_per_block, double *__restrict__ partial)
{
    const BnWalk w = bn_walk<VW>(rows, C, rows_per_block);
    double s[VW], q[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) s[j] = q[j] = 0.0;
    if (w.col >= 0) {
        const float *p = z + w.col;
        i64 r = w.r0;
        // four rows in flight per thread
        for (; r + 3 * (i64)w.rstep < w.rend; r += 4 * (i64)w.rstep) {
            float v[4][VW];
#pragma unroll
            for (int u = 0; u < 4; ++u) bn_load<VW>(p + (r + (i64)u * w.rstep) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const double d = (double)v[u][j];
                    s[j] += d;
                    q[j] += d * d;
                }
        }
        for (; r < w.rend; r += w.rstep) {
            float v[VW];
            bn_load<VW>(p + r * C, v);
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const double d = (double)v[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
    }
    bn_block_fold<VW>(s, q, C, partial);
}

// Column sums of g and g * zhat, with zhat = (z - mean) * invstd, u = zhat * gamma + beta and
// g = grad_y * (u > 0 ? 1 : slope) (act) or grad_y.
template <int VW>
__global__ __launch_bounds__(BN_THREADS) void bn_grad_sums_kernel(
    const float *__restrict__ gy, const float *__restrict__ z, i64 rows, int C, i64 rows_per_block,
    const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
    const float *__restrict__ weight, const float *__restrict__ bias, int act, float slope,
    double *__restrict__ partial)
{
    const BnWalk w = bn_walk<VW>(rows, C, rows_per_block);
    double s[VW], q[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) s[j] = q[j] = 0.0;
    if (w.col >= 0) {
        BnTerms<VW> T;
        T.load(w.col, save_mean, save_invstd, weight, bias);
        i64 r = w.r0;
        for (; r + (i64)w.rstep < w.rend; r += 2 * (i64)w.rstep) {
            float v[2][VW], g[2][VW];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bn_load<VW>(z + (r + (i64)u * w.rstep) * C + w.col, v[u]);
                bn_load<VW>(gy + (r + (i64)u * w.rstep) * C + w.col, g[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const float zh = (v[u][j] - T.mean[j]) * T.invstd[j];
                    const float act_in = zh * T.gamma[j] + T.beta[j];
                    const float gg = (act && !(act_in > 0.0f)) ? g[u][j] * slope : g[u][j];
                    s[j] += (double)gg;
                    q[j] += (double)gg * (double)zh;
                }
        }
        for (; r < w.rend; r += w.rstep) {
            float v[VW], g[VW];
            bn_load<VW>(z + r * C + w.col, v);
            bn_load<VW>(gy + r * C + w.col, g);
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const float zh = (v[j] - T.mean[j]) * T.invstd[j];
                const float act_in = zh * T.gamma[j] + T.beta[j];
                const float gg = (act && !(act_in > 0.0f)) ? g[j] * slope : g[j];
                s[j] += (double)gg;
                q[j] += (double)gg * (double)zh;
            }
        }
    }
    bn_block_fold<VW>(s, q, C, partial);
}

// Fold of the partial rows: BN_FOLD_COLS columns per block, each by BN_THREADS / BN_FOLD_COLS
// threads that take interleaved partial rows; their sums are added pairwise through LDS (a
// fixed tree).  out0 / out1: the two column sums, valid in the thread with owner set.
__device__ __forceinline__ void bn_fold(const double *partial, int nblk, int C, double &out0, double &out1, bool &owner, int &col)
{
    __shared__ double red[BN_THREADS][2];
    const int t = threadIdx.x;
    const int c = t % BN_FOLD_COLS, slice = t / BN_FOLD_COLS;
    constexpr int SLICES = BN_THREADS / BN_FOLD_COLS;
    col = blockIdx.x * BN_FOLD_COLS + c;
    double a = 0.0, b = 0.0;
    if (col < C)
        for (int k = slice; k < nblk; k += SLICES) {
            a += partial[(size_t)k * 2 * C + col];
            b += partial[(size_t)k * 2 * C + C + col];
        }
    red[t][0] = a;
    red[t][1] = b;
    __syncthreads();
    for (int half = SLICES / 2; half >= 1; half >>= 1) {
        if (slice < half) {
            red[t][0] += red[t + half * BN_FOLD_COLS][0];
            red[t][1] += red[t + half * BN_FOLD_COLS][1];
        }
        __syncthreads();
    }
    owner = slice == 0 && col < C;
    out0 = red[t][0];
    out1 = red[t][1];
}

// Batch statistics from the column sums; running statistics updated as nn.BatchNorm1d does
// (biased variance for the normalisation, unbiased for the running estimate).
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(
    const double *__restrict__ partial, int nblk, int C, i64 rows, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var,
    float *__restrict__ save_mean, float *__restrict__ save_invstd)
{
    double S, Q;
    bool owner;
    int col;
    bn_fold(partial, nblk, C, S, Q, owner, col);
    if (!owner) return;
    const double n = (double)rows;
    const double mean = S / n;
    double var = Q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[col] = (float)mean;
    save_invstd[col] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[col] = (float)((1.0 - (double)momentum) * (double)running_mean[col] + (double)momentum * mean);
    if (running_var) {
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_var[col] = (float)((1.0 - (double)momentum) * (double)running_var[col] + (double)momentum * unbiased);
    }
}

// Normalisation terms from the running statistics (eval mode with a graph).
__global__ void bn_running_terms_kernel(
    const float *__restrict__ running_mean, const float *__restrict__ running_var, int C, float eps,
    float *__restrict__ save_mean, float *__restrict__ save_invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = running_mean[c];
    save_invstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
}

// Gradients of weight and bias, and the two per-column means the input gradient needs.
__global__ __launch_bounds__(BN_THREADS) void bn_grad_terms_kernel(
    const double *__restrict__ partial, int nblk, int C, i64 rows,
    float *__restrict__ grad_weight, float *__restrict__ grad_bias, float *__restrict__ col_terms)
{
    double S, Q;
    bool owner;
    int col;
    bn_fold(partial, nblk, C, S, Q, owner, col);
    if (!owner) return;
    if (grad_bias) grad_bias[col] = (float)S;
    if (grad_weight) grad_weight[col] = (float)Q;
    col_terms[col] = (float)(S / (double)rows);
    col_terms[C + col] = (float)(Q / (double)rows);
}

// y = act((z - mean) * invstd * gamma + beta)
template <int VW>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(
    const float *__restrict__ z, i64 rows, int C, i64 rows_per_block,
    const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
    const float *__restrict__ weight, const float *__restrict__ bias, int act, float slope,
    float *__restrict__ y)
{
    const BnWalk w = bn_walk<VW>(rows, C, rows_per_block);
    if (w.col < 0) return;
    BnTerms<VW> T;
    T.load(w.col, save_mean, save_invstd, weight, bias);
    i64 r = w.r0;
    for (; r + 3 * (i64)w.rstep < w.rend; r += 4 * (i64)w.rstep) {
        float v[4][VW];
#pragma unroll
        for (int u = 0; u < 4; ++u) bn_load<VW>(z + (r + (i64)u * w.rstep) * C + w.col, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const float a = (v[u][j] - T.mean[j]) * T.invstd[j] * T.gamma[j] + T.beta[j];
                v[u][j] = (act && !(a > 0.0f)) ? a * slope : a;
            }
            bn_store<VW>(y + (r + (i64)u * w.rstep) * C + w.col, v[u]);
        }
    }
    for (; r < w.rend; r += w.rstep) {
        float v[VW];
        bn_load<VW>(z + r * C + w.col, v);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float a = (v[j] - T.mean[j]) * T.invstd[j] * T.gamma[j] + T.beta[j];
            v[j] = (act && !(a > 0.0f)) ? a * slope : a;
        }
        bn_store<VW>(y + r * C + w.col, v);
    }
}

// grad_z = gamma * invstd * (g - mean(g) - zhat * mean(g * zhat))   (batch statistics)
//        = gamma * invstd * g                                        (running statistics)
template <int VW>
__global__ __launch_bounds__(BN_THREADS) void bn_grad_apply_kernel(
    const float *__restrict__ gy, const float *__restrict__ z, i64 rows, int C, i64 rows_per_block,
    const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
    const float *__restrict__ weight, const float *__restrict__ bias, int act, float slope,
    const float *__restrict__ col_terms, int training, float *__restrict__ gz)
{
    const BnWalk w = bn_walk<VW>(rows, C, rows_per_block);
    if (w.col < 0) return;
    BnTerms<VW> T;
    T.load(w.col, save_mean, save_invstd, weight, bias);
    float m_g[VW], m_gz[VW], k[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        m_g[j] = training ? col_terms[w.col + j] : 0.0f;
        m_gz[j] = training ? col_terms[C + w.col + j] : 0.0f;
        k[j] = T.gamma[j] * T.invstd[j];
    }
    i64 r = w.r0;
    for (; r + (i64)w.rstep < w.rend; r += 2 * (i64)w.rstep) {
        float v[2][VW], g[2][VW];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bn_load<VW>(z + (r + (i64)u * w.rstep) * C + w.col, v[u]);
            bn_load<VW>(gy + (r + (i64)u * w.rstep) * C + w.col, g[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const float zh = (v[u][j] - T.mean[j]) * T.invstd[j];
                const float act_in = zh * T.gamma[j] + T.beta[j];
                const float gg = (act && !(act_in > 0.0f)) ? g[u][j] * slope : g[u][j];
                v[u][j] = (gg - m_g[j] - zh * m_gz[j]) * k[j];
            }
            bn_store<VW>(gz + (r + (i64)u * w.rstep) * C + w.col, v[u]);
        }
    }
    for (; r < w.rend; r += w.rstep) {
        float v[VW], g[VW];
        bn_load<VW>(z + r * C + w.col, v);
        bn_load<VW>(gy + r * C + w.col, g);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float zh = (v[j] - T.mean[j]) * T.invstd[j];
            const float act_in = zh * T.gamma[j] + T.beta[j];
            const float gg = (act && !(act_in > 0.0f)) ? g[j] * slope : g[j];
            v[j] = (gg - m_g[j] - zh * m_gz[j]) * k[j];
        }
        bn_store<VW>(gz + r * C + w.col, v);
    }
}

// Launch shape of the row-walking kernels: (row blocks, column chunks), rows per block a
// multiple of the rows a block covers per pass.
struct BnGrid {
    dim3 grid;
    i64 rows_per_block;
    int nblk;
};

static BnGrid bn_grid(i64 rows, int C, int VW)
{
    const int CV = C / VW;
    const int cw = CV < BN_THREADS ? CV : BN_THREADS;
    const int rpp = BN_THREADS / cw;
    const int chunks = (CV + BN_THREADS - 1) / BN_THREADS;
    // at least eight passes per block, at most BN_MAX_BLOCKS blocks
    i64 nblk = (rows + (i64)rpp * 8 - 1) / ((i64)rpp * 8);
    if (nblk > BN_MAX_BLOCKS) nblk = BN_MAX_BLOCKS;
    if (nblk < 1) nblk = 1;
    i64 rpb = (rows + nblk - 1) / nblk;
    rpb = (rpb + rpp - 1) / rpp * rpp;
    nblk = (rows + rpb - 1) / rpb;
    if (nblk < 1) nblk = 1;
    BnGrid g;
    g.grid = dim3((unsigned)nblk, (unsigned)chunks);
    g.rows_per_block = rpb;
    g.nblk = (int)nblk;
    return g;
}

static size_t bn_partial_bytes(int C) { return align_up((size_t)BN_MAX_BLOCKS * 2 * (size_t)C * sizeof(double)); }

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_batch_norm_act_workspace_bytes(int32_t C)
{
    if (C < 1) return 0;
    return bn_partial_bytes(C) + align_up(2 * (size_t)C * sizeof(float));
}

extern "C" int mvf_batch_norm_act_forward(
    const float *z, int64_t rows, int32_t C, const float *weight, const float *bias,
    float *running_mean, float *running_var, int32_t training, float momentum, float eps,
    int32_t act, float slope, float *y, float *save_mean, float *save_invstd,
    const double *col_sums, int32_t n_col_sums, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(rows >= 1 && C >= 1, "batch_norm_act_forward: rows and C must be positive");
    MVF_REQUIRE(z && y && save_mean && save_invstd, "batch_norm_act_forward: null argument");
    MVF_REQUIRE(training || (running_mean && running_var),
                "batch_norm_act_forward: running statistics are needed outside train mode");
    MVF_REQUIRE(!training || rows > 1, "batch_norm_act_forward: more than one row per channel is needed in train mode");
    MVF_REQUIRE(!training || col_sums || (workspace && workspace_bytes >= mvf_batch_norm_act_workspace_bytes(C)),
                "batch_norm_act_forward: workspace too small");
    MVF_REQUIRE(!col_sums || n_col_sums >= 1, "batch_norm_act_forward: column sums without partial rows");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)y % 16 == 0;
    const BnGrid g = bn_grid(rows, C, vec ? 4 : 1);
    if (training) {
        const double *partial = col_sums;
        int nblk = n_col_sums;
        if (!col_sums) {
            double *mine = (double *)workspace;
            if (vec) hipLaunchKernelGGL(bn_sums_kernel<4>, g.grid, dim3(BN_THREADS), 0, st, z, (i64)rows, C, g.rows_per_block, mine);
            else hipLaunchKernelGGL(bn_sums_kernel<1>, g.grid, dim3(BN_THREADS), 0, st, z, (i64)rows, C, g.rows_per_block, mine);
            partial = mine;
            nblk = g.nblk;
        }
        hipLaunchKernelGGL(bn_stats_kernel, dim3(cdiv(C, BN_FOLD_COLS)), dim3(BN_THREADS), 0, st,
                           partial, nblk, C, (i64)rows, eps, momentum,
                           running_mean, running_var, save_mean, save_invstd);
    } else {
        hipLaunchKernelGGL(bn_running_terms_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st,
                           (const float *)running_mean, (const float *)running_var, C, eps, save_mean, save_invstd);
    }
    if (vec)
        hipLaunchKernelGGL(bn_apply_kernel<4>, g.grid, dim3(BN_THREADS), 0, st, z, (i64)rows, C, g.rows_per_block,
                           (const float *)save_mean, (const float *)save_invstd, weight, bias, act, slope, y);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, g.grid, dim3(BN_THREADS), 0, st, z, (i64)rows, C, g.rows_per_block,
                           (const float *)save_mean, (const float *)save_invstd, weight, bias, act, slope, y);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_batch_norm_act_backward(
    const float *grad_y, const float *z, int64_t rows, int32_t C, const float *weight, const float *bias,
    const float *save_mean, const float *save_invstd, int32_t training, int32_t act, float slope,
    float *grad_z, float *grad_weight, float *grad_bias,
    void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(rows >= 1 && C >= 1, "batch_norm_act_backward: rows and C must be positive");
    MVF_REQUIRE(grad_y && z && save_mean && save_invstd && grad_z, "batch_norm_act_backward: null argument");
    MVF_REQUIRE(workspace && workspace_bytes >= mvf_batch_norm_act_workspace_bytes(C),
                "batch_norm_act_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)grad_y % 16 == 0 && (uintptr_t)grad_z % 16 == 0;
    const BnGrid g = bn_grid(rows, C, vec ? 4 : 1);
    double *partial = (double *)workspace;
    float *col_terms = (float *)((char *)workspace + bn_partial_bytes(C));
    if (training || grad_weight || grad_bias) {
        if (vec)
            hipLaunchKernelGGL(bn_grad_sums_kernel<4>, g.grid, dim3(BN_THREADS), 0, st, grad_y, z, (i64)rows, C,
                               g.rows_per_block, save_mean, save_invstd, weight, bias, act, slope, partial);
        else
            hipLaunchKernelGGL(bn_grad_sums_kernel<1>, g.grid, dim3(BN_THREADS), 0, st, grad_y, z, (i64)rows, C,
                               g.rows_per_block, save_mean, save_invstd, weight, bias, act, slope, partial);
        hipLaunchKernelGGL(bn_grad_terms_kernel, dim3(cdiv(C, BN_FOLD_COLS)), dim3(BN_THREADS), 0, st,
                           (const double *)partial, g.nblk, C, (i64)rows, grad_weight, grad_bias, col_terms);
    }
    if (vec)
        hipLaunchKernelGGL(bn_grad_apply_kernel<4>, g.grid, dim3(BN_THREADS), 0, st, grad_y, z, (i64)rows, C,
                           g.rows_per_block, save_mean, save_invstd, weight, bias, act, slope,
                           (const float *)col_terms, training, grad_z);
    else
        hipLaunchKernelGGL(bn_grad_apply_kernel<1>, g.grid, dim3(BN_THREADS), 0, st, grad_y, z, (i64)rows, C,
                           g.rows_per_block, save_mean, save_invstd, weight, bias, act, slope,
                           (const float *)col_terms, training, grad_z);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
