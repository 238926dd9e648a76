// Dense pieces of the attentive pools (modules/multimodal/pooling.py:159-551,
// 604-727): chains of Linear + eval-mode BatchNorm + LeakyReLU layers on the
// fp32 matrix cores, the query/key softmax and the attention-weighted segment
// sum.
//
// MLP chain.  A wave owns a tile of 32 rows; its activations live in a private
// LDS tile between layers, so a chain of up to three layers reads its input
// once from HBM and writes its output once.  Every layer is a sequence of
// v_mfma_f32_32x32x2_f32: A = the activation tile (lane l holds row l & 31,
// k = l >> 5), B = the transposed weights in LDS (lane l holds k = l >> 5,
// output column l & 31), 16 accumulator registers per 32 output columns
// (column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)).  The instruction is
// an exact k-ordered fp32 fmaf chain, so results do not depend on the tiling.
// BatchNorm (eval) and the bias collapse into one scale and one shift per
// output column, applied with the activation in the epilogue.
//
// Limits of one launch: every layer's output width <= 64 and every layer
// after the first has input width <= 64; the first layer's input is streamed
// in chunks of 64 columns and may be any width.  The host splits wider layers
// into several launches.
#include "mvf_common.h"

namespace mvf {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MLP_THREADS = 256;
constexpr int MLP_WAVES = 4;
constexpr int MLP_ROWS = 32 * MLP_WAVES;   // rows per workgroup step
constexpr int MLP_KC = 64;                 // input columns per chunk
constexpr int MLP_PITCH = 66;              // floats per tile row: lanes (row, k) hit distinct banks
constexpr int MLP_MAX_LAYERS = 3;
constexpr int MLP_MAX_C = 64;

struct MlpSrc {
    const float *p;       // [*, ld] row-major, null = absent
    const int *gather;    // row -> source row, null = identity
    int ld, cols;
};

struct MlpArgs {
    MlpSrc a, b;          // input row = [a | b] (mode 0), a + b (mode 1), [a | a + b] (mode 2)
    int mode;
    int nlayers;
    i64 rows;
    mvf_mlp_layer L[MLP_MAX_LAYERS];
    float *out;
    int ldo;
};

__device__ __forceinline__ int pad2(int v) { return (v + 1) & ~1; }
__device__ __forceinline__ int pad32(int v) { return (v + 31) & ~31; }

// Orders LDS traffic between the lanes of one wave (tile writes before reads).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float mlp_input(const MlpArgs &A, i64 ra, i64 rb, int c)
{
    const int ca = A.a.cols;
    if (A.mode == 1) return __fadd_rn(A.a.p[ra * A.a.ld + c], A.b.p[rb * A.b.ld + c]);
    if (c < ca) return A.a.p[ra * A.a.ld + c];
    c -= ca;
    if (A.mode == 2) return __fadd_rn(A.a.p[ra * A.a.ld + c], A.b.p[rb * A.b.ld + c]);
    return A.b.p[rb * A.b.ld + c];
}

// One layer's K loop over kc2 / 2 steps: acc += tile[32 x kc2] * wt[kc2 x cop].
__device__ __forceinline__ void mlp_mma(
    const float *tile, const float *wt, int kc2, int cop, int lane, floatx16 &acc0, floatx16 &acc1)
{
    const int i = lane & 31, kh = lane >> 5;
    const float *ta = tile + i * MLP_PITCH + kh;
    const float *wb = wt + kh * cop + i;
    if (cop > 32) {
        for (int k = 0; k < kc2; k += 2) {
            const float a = ta[k];
            const float b0 = wb[k * cop], b1 = wb[k * cop + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < kc2; k += 2) {
            const float a = ta[k];
            const float b0 = wb[k * cop];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
        }
    }
}

// Scale, shift, activation; to the wave's tile (next layer's input) or to out.
__device__ __forceinline__ void mlp_epilogue(
    const floatx16 &acc, int col0, const float *sc, const float *sh, const mvf_mlp_layer &L,
    float *tile, float *out, int ldo, i64 row0, i64 rows, int lane)
{
    const int j = col0 + (lane & 31), kh = lane >> 5;
    const float s = sc[j], t = sh[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = __fadd_rn(__fmul_rn(acc[r], s), t);
        if (L.act) v = v > 0.f ? v : __fmul_rn(v, L.slope);
        if (out) {
            if (row0 + row < rows && j < L.cout) out[(row0 + row) * ldo + j] = v;
        } else {
            tile[row * MLP_PITCH + j] = v;
        }
    }
}

__global__ __launch_bounds__(MLP_THREADS) void mlp_chain_kernel(MlpArgs A)
{
    extern __shared__ float smem[];
    float *sc = smem;                                   // [3][64] column scales
    float *sh = sc + MLP_MAX_LAYERS * MLP_MAX_C;        // [3][64] column shifts
    float *tiles = sh + MLP_MAX_LAYERS * MLP_MAX_C;     // one 32-row tile per wave
    float *wt[MLP_MAX_LAYERS];                          // transposed weights [k][cout_pad]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float *tile = tiles + wave * 32 * MLP_PITCH;

    const int cin0 = A.L[0].cin;
    const int nchunks = (cin0 + MLP_KC - 1) / MLP_KC;
    {
        float *w = tiles + MLP_WAVES * 32 * MLP_PITCH;
        for (int l = 0; l < A.nlayers; ++l) {
            wt[l] = w;
            const int kin = l == 0 ? (cin0 < MLP_KC ? pad2(cin0) : MLP_KC) : pad2(A.L[l].cin);
            w += kin * pad32(A.L[l].cout);
        }
    }
    // column scale / shift: y = (x W^T + bias - mean) / sqrt(var + eps) * gamma + beta
    for (int idx = tid; idx < A.nlayers * MLP_MAX_C; idx += MLP_THREADS) {
        const int l = idx / MLP_MAX_C, j = idx - l * MLP_MAX_C;
        const mvf_mlp_layer &L = A.L[l];
        float s = 0.f, t = 0.f;
        if (j < L.cout) {
            s = 1.f;
            if (L.bn_weight) {
                s = __fdiv_rn(L.bn_weight[j], sqrtf(__fadd_rn(L.bn_var[j], L.bn_eps)));
                t = __fsub_rn(L.bn_bias[j], __fmul_rn(L.bn_mean[j], s));
            }
            if (L.bias) t = __fadd_rn(__fmul_rn(L.bias[j], s), t);
        }
        sc[idx] = s;
        sh[idx] = t;
    }
    // resident weights: every layer but a chunked first one
    for (int l = (nchunks > 1 ? 1 : 0); l < A.nlayers; ++l) {
        const mvf_mlp_layer &L = A.L[l];
        const int kin = pad2(L.cin), cop = pad32(L.cout);
        for (int idx = tid; idx < kin * cop; idx += MLP_THREADS) {
            const int k = idx / cop, j = idx - k * cop;
            wt[l][idx] = (k < L.cin && j < L.cout) ? L.weight[(size_t)j * L.cin + k] : 0.f;
        }
    }
    __syncthreads();

    const bool gather_a = A.a.gather != nullptr, gather_b = A.b.gather != nullptr;
    for (i64 base = (i64)blockIdx.x * MLP_ROWS; base < A.rows; base += (i64)gridDim.x * MLP_ROWS) {
        const i64 row0 = base + wave * 32;
        floatx16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

        // ---- first layer, input streamed in chunks of 64 columns ----
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int k0 = chunk * MLP_KC;
            const int kc = cin0 - k0 < MLP_KC ? cin0 - k0 : MLP_KC;
            const int kc2 = pad2(kc), cop = pad32(A.L[0].cout);
            if (nchunks > 1) {
                __syncthreads();                      // every wave is done with the previous chunk
                const mvf_mlp_layer &L = A.L[0];
                for (int idx = tid; idx < kc2 * cop; idx += MLP_THREADS) {
                    const int k = idx / cop, j = idx - k * cop;
                    wt[0][idx] = (k < kc && j < L.cout) ? L.weight[(size_t)j * L.cin + k0 + k] : 0.f;
                }
            }
            // the wave's 32 x 64 input tile, one row per step, zero beyond the data
#pragma unroll 4
            for (int r = 0; r < 32; ++r) {
                const i64 row = row0 + r;
                float v = 0.f;
                if (row < A.rows && lane < kc) {
                    const i64 ra = gather_a ? (i64)A.a.gather[row] : row;
                    const i64 rb = gather_b ? (i64)A.b.gather[row] : row;
                    v = mlp_input(A, ra, rb, k0 + lane);
                }
                tile[r * MLP_PITCH + lane] = v;
            }
            if (nchunks > 1) __syncthreads(); else wave_sync();
            mlp_mma(tile, wt[0], kc2, cop, lane, acc0, acc1);
            wave_sync();
        }
        // ---- epilogues and the following layers ----
        for (int l = 0; l < A.nlayers; ++l) {
            const mvf_mlp_layer &L = A.L[l];
            const bool last = l == A.nlayers - 1;
            float *o = last ? A.out : nullptr;
            const int cop = pad32(L.cout);
            mlp_epilogue(acc0, 0, sc + l * MLP_MAX_C, sh + l * MLP_MAX_C, L, tile, o, A.ldo, row0, A.rows, lane);
            if (cop > 32)
                mlp_epilogue(acc1, 32, sc + l * MLP_MAX_C, sh + l * MLP_MAX_C, L, tile, o, A.ldo, row0, A.rows, lane);
            if (last) break;
            // the tile's pad column feeds the next layer's odd k: keep it finite
            if ((L.cout & 1) && lane < 32) tile[lane * MLP_PITCH + L.cout] = 0.f;
            wave_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            mlp_mma(tile, wt[l + 1], pad2(A.L[l + 1].cin), pad32(A.L[l + 1].cout), lane, acc0, acc1);
            wave_sync();
        }
    }
}

// ------------------------------------------------------------------------- //
// attention scores inside CSR segments                                       //
// ------------------------------------------------------------------------- //
// One thread per (segment, group).  comp[r][g] = sum_k K[r][g D + k] Q[s][g D + k]
// (products and sums rounded one by one, k ascending; optionally / sqrt(D)), or
// comp is given.  att = segment softmax of comp (max-centred, optional
// / sqrt(rows), eps in the denominator; pooling.py:770-822), gate[s][g] =
// tanh(relu(max_r comp * Gw[g] + Gb[g])) (pooling.py:717-723, 523-526).
__global__ __launch_bounds__(256) void attention_scores_kernel(
    const float *__restrict__ keys, int ldk, const float *__restrict__ queries, int ldq,
    int q_per_row, const i64 *__restrict__ ptr, i64 nseg, int G, int D, float dim_div, int group_scaling,
    float eps, const float *__restrict__ gate_w, const float *__restrict__ gate_b,
    float *__restrict__ comp, float *__restrict__ att, float *__restrict__ gate)
{
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= nseg * G) return;
    const i64 s = t / G;
    const int g = (int)(t - s * G);
    const i64 a = ptr[s], b = ptr[s + 1];
    float mx = 0.f;                                    // segment max of nothing is 0
    if (keys) {
        for (i64 r = a; r < b; ++r) {
            const float *q = queries + (size_t)(q_per_row ? r : s) * ldq + (size_t)g * D;
            const float *k = keys + (size_t)r * ldk + (size_t)g * D;
            float c = __fmul_rn(k[0], q[0]);
            for (int d = 1; d < D; ++d) c = __fadd_rn(c, __fmul_rn(k[d], q[d]));
            if (dim_div != 0.f) c = __fdiv_rn(c, dim_div);
            comp[(size_t)r * G + g] = c;
            mx = (r == a || c > mx) ? c : mx;
        }
    } else {
        for (i64 r = a; r < b; ++r) {
            const float c = comp[(size_t)r * G + g];
            mx = (r == a || c > mx) ? c : mx;
        }
    }
    if (gate) {
        float x = mx;
        if (gate_w) x = __fmul_rn(x, gate_w[g]);
        if (gate_b) x = __fadd_rn(x, gate_b[g]);
        gate[t] = tanhf(x > 0.f ? x : 0.f);
    }
    if (b <= a) return;
    const float sq = sqrtf((float)(b - a));
    float sum = 0.f;
    for (i64 r = a; r < b; ++r) {
        float c = __fsub_rn(comp[(size_t)r * G + g], mx);
        if (group_scaling) c = __fdiv_rn(c, sq);
        const float e = expf(c);
        att[(size_t)r * G + g] = e;
        sum = (r == a) ? e : __fadd_rn(sum, e);
    }
    sum = __fadd_rn(sum, eps);
    for (i64 r = a; r < b; ++r) att[(size_t)r * G + g] = __fdiv_rn(att[(size_t)r * G + g], sum);
}

// x_pool[s][c] = (sum_r x[r][c] * att[r][group(c)]) * gate[s][group(c)], rows
// in order, one (segment, channel) pair per thread step as in segment_csr.
__global__ __launch_bounds__(256) void attention_pool_kernel(
    const float *__restrict__ x, int C, const float *__restrict__ att, int G,
    const int *__restrict__ chan2group, const float *__restrict__ gate,
    const i64 *__restrict__ ptr, i64 nseg, float *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= nseg * C) return;
    const i64 s = e / C;
    const int c = (int)(e - s * C);
    const int g = chan2group ? chan2group[c] : 0;
    const i64 a = ptr[s], b = ptr[s + 1];
    float acc = 0.f;
    if (b > a) {
        acc = __fmul_rn(x[(size_t)a * C + c], att[(size_t)a * G + g]);
        for (i64 r = a + 1; r < b; ++r)
            acc = __fadd_rn(acc, __fmul_rn(x[(size_t)r * C + c], att[(size_t)r * G + g]));
    }
    if (gate) acc = __fmul_rn(acc, gate[(size_t)s * G + g]);
    out[e] = acc;
}

// row -> segment index of a CSR pointer array.
__global__ __launch_bounds__(256) void csr_row_segments_kernel(
    const i64 *__restrict__ ptr, i64 nseg, int *__restrict__ row2seg)
{
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const i64 a = ptr[s], b = ptr[s + 1];
    for (i64 r = a; r < b; ++r) row2seg[r] = (int)s;
}

// HeuristicBimodalCSRPool (pooling.py:129-152): per segment the row with the
// largest / smallest value of one mapping feature (first one on ties), its
// modality features copied out; zeros for an empty segment.
__global__ __launch_bounds__(256) void segment_pick_kernel(
    const float *__restrict__ x_mod, int C, const float *__restrict__ feat, int ldf,
    const i64 *__restrict__ ptr, i64 nseg, int pick_max, float *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= nseg * C) return;
    const i64 s = e / C;
    const int c = (int)(e - s * C);
    const i64 a = ptr[s], b = ptr[s + 1];
    float v = 0.f;
    if (b > a) {
        i64 best = a;
        float fb = feat[(size_t)a * ldf];
        for (i64 r = a + 1; r < b; ++r) {
            const float f = feat[(size_t)r * ldf];
            if (pick_max ? f > fb : f < fb) { fb = f; best = r; }
        }
        v = x_mod[(size_t)best * C + c];
    }
    out[e] = v;
}

static size_t mlp_lds_bytes(const mvf_mlp_layer *L, int n)
{
    size_t f = 2 * MLP_MAX_LAYERS * MLP_MAX_C + (size_t)MLP_WAVES * 32 * MLP_PITCH;
    for (int l = 0; l < n; ++l) {
        const int kin = l == 0 ? (L[l].cin < MLP_KC ? ((L[l].cin + 1) & ~1) : MLP_KC) : ((L[l].cin + 1) & ~1);
        f += (size_t)kin * ((L[l].cout + 31) & ~31);
    }
    return f * sizeof(float);
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_mlp_forward(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    float *out, int32_t ldo, void *stream)
{
    MVF_REQUIRE(rows >= 0 && layers && nlayers >= 1 && nlayers <= MLP_MAX_LAYERS,
                "mlp_forward: bad arguments (nlayers=%d)", nlayers);
    MVF_REQUIRE(mode >= 0 && mode <= 2, "mlp_forward: unknown input mode %d", mode);
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(xa && cols_a >= 1 && lda >= cols_a && out, "mlp_forward: bad first input");
    MVF_REQUIRE((mode == 0 && !xb) || (xb && cols_b >= 1 && ldb >= cols_b),
                "mlp_forward: bad second input");
    MVF_REQUIRE(mode == 0 || cols_a == cols_b, "mlp_forward: summed inputs differ in width");
    const int cin0 = mode == 0 ? cols_a + (xb ? cols_b : 0) : (mode == 1 ? cols_a : 2 * cols_a);
    MVF_REQUIRE(layers[0].cin == cin0, "mlp_forward: first layer takes %d columns, input has %d",
                layers[0].cin, cin0);
    for (int l = 0; l < nlayers; ++l) {
        MVF_REQUIRE(layers[l].weight && layers[l].cin >= 1 && layers[l].cout >= 1 &&
                    layers[l].cout <= MLP_MAX_C, "mlp_forward: layer %d width %d -> %d unsupported",
                    l, layers[l].cin, layers[l].cout);
        MVF_REQUIRE(l == 0 || (layers[l].cin == layers[l - 1].cout && layers[l].cin <= MLP_MAX_C),
                    "mlp_forward: layer %d input width %d does not chain", l, layers[l].cin);
        MVF_REQUIRE(!layers[l].bn_weight || (layers[l].bn_bias && layers[l].bn_mean && layers[l].bn_var),
                    "mlp_forward: layer %d has partial batch-norm parameters", l);
    }
    MVF_REQUIRE(ldo >= layers[nlayers - 1].cout, "mlp_forward: output stride too small");
    MlpArgs A;
    A.a = MlpSrc{xa, gather_a, lda, cols_a};
    A.b = MlpSrc{xb, gather_b, ldb, xb ? cols_b : 0};
    A.mode = mode;
    A.nlayers = nlayers;
    A.rows = rows;
    for (int l = 0; l < nlayers; ++l) A.L[l] = layers[l];
    A.out = out;
    A.ldo = ldo;
    const size_t lds = mlp_lds_bytes(layers, nlayers);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const i64 want = (rows + MLP_ROWS - 1) / MLP_ROWS, cap = (i64)cus * 2;
    dim3 grid((unsigned)(want < cap ? want : cap));
    MVF_HIP(hipFuncSetAttribute((const void *)mlp_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(mlp_chain_kernel, grid, dim3(MLP_THREADS), lds, (hipStream_t)stream, A);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_attention_pool(
    const float *keys, int32_t ldk, const float *queries, int32_t ldq, int32_t q_per_row,
    const float *x_mod, int32_t C, const int64_t *ptr, int64_t nseg, int64_t rows, int32_t G, int32_t D,
    int32_t dim_scaling, int32_t group_scaling, float eps, const int32_t *chan2group,
    const float *gate_w, const float *gate_b, int32_t gating,
    float *comp, float *att, float *gate, float *out, void *stream)
{
    MVF_REQUIRE(nseg >= 0 && rows >= 0 && G >= 1 && C >= 1, "attention_pool: bad shape");
    MVF_REQUIRE(G == 1 || chan2group, "attention_pool: groups need the channel map");
    MVF_REQUIRE(!keys || (queries && D >= 1 && ldk >= G * D && ldq >= G * D),
                "attention_pool: bad key / query layout");
    MVF_REQUIRE(!gating || gate, "attention_pool: gating needs the gate buffer");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(ptr && out && (rows == 0 || (x_mod && comp && att)), "attention_pool: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const float dim_div = (keys && dim_scaling) ? (float)sqrt((double)D) : 0.f;
    hipLaunchKernelGGL(attention_scores_kernel, dim3(cdiv(nseg * G, 256)), dim3(256), 0, st,
                       keys, ldk, queries, ldq, q_per_row, (const i64 *)ptr, (i64)nseg, G, D, dim_div,
                       group_scaling, eps, gate_w, gate_b, comp, att, gating ? gate : (float *)nullptr);
    hipLaunchKernelGGL(attention_pool_kernel, dim3(cdiv(nseg * C, 256)), dim3(256), 0, st,
                       x_mod, C, att, G, G == 1 ? (const int *)nullptr : chan2group,
                       gating ? gate : (const float *)nullptr, (const i64 *)ptr, (i64)nseg, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_csr_row_segments(const int64_t *ptr, int64_t nseg, int32_t *row2seg, void *stream)
{
    MVF_REQUIRE(nseg >= 0, "csr_row_segments: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(ptr && row2seg, "csr_row_segments: null buffer");
    hipLaunchKernelGGL(csr_row_segments_kernel, dim3(cdiv(nseg, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const i64 *)ptr, (i64)nseg, row2seg);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_segment_pick(
    const float *x_mod, int32_t C, const float *feat, int32_t ldf, const int64_t *ptr,
    int64_t nseg, int32_t pick_max, float *out, void *stream)
{
    MVF_REQUIRE(nseg >= 0 && C >= 1 && ldf >= 1, "segment_pick: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(x_mod && feat && ptr && out, "segment_pick: null buffer");
    hipLaunchKernelGGL(segment_pick_kernel, dim3(cdiv(nseg * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x_mod, C, feat, ldf, (const i64 *)ptr, (i64)nseg,
                       pick_max, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
