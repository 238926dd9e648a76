// CSR segment kernels: the HIP replacement of torch_scatter.segment_csr and of
// the gather / softmax helpers of modules/multimodal/pooling.py.
//
// Layout: src is [rows][C] float32 row-major, ptr is int64 [nseg+1].  The
// output is walked as a flat array of nseg * (C / VEC) vectors: a thread owns
// one (segment, column) pair at a time and folds the segment's rows in order,
// so sums are identical from run to run and identical to a sequential CPU
// fold.  Consecutive threads own consecutive output vectors (and, for the
// short segments of the view pools, consecutive input vectors); every thread
// keeps four independent pairs in flight because segments are short and the
// chain ptr -> row -> store is latency-bound otherwise.  HBM-bound: rows*C*4
// bytes read once, nseg*C*4 bytes written once.
#include "mvf_common.h"

namespace mvf {

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<4> { typedef float4 T; };

template <int MODE> __device__ __forceinline__ float fold(float a, float x)
{
    if (MODE == MVF_REDUCE_MAX) return x > a ? x : a;
    if (MODE == MVF_REDUCE_MIN) return x < a ? x : a;
    return __fadd_rn(a, x);
}

template <int MODE> __device__ __forceinline__ void fold_v(float &a, const float &x) { a = fold<MODE>(a, x); }
template <int MODE> __device__ __forceinline__ void fold_v(float4 &a, const float4 &x)
{
    a.x = fold<MODE>(a.x, x.x); a.y = fold<MODE>(a.y, x.y);
    a.z = fold<MODE>(a.z, x.z); a.w = fold<MODE>(a.w, x.w);
}
__device__ __forceinline__ void scale_v(float &a, float c) { a = __fdiv_rn(a, c); }
__device__ __forceinline__ void scale_v(float4 &a, float c)
{
    a.x = __fdiv_rn(a.x, c); a.y = __fdiv_rn(a.y, c); a.z = __fdiv_rn(a.z, c); a.w = __fdiv_rn(a.w, c);
}
__device__ __forceinline__ void zero_v(float &a) { a = 0.f; }
__device__ __forceinline__ void zero_v(float4 &a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

constexpr int SEG_ILP = 4;

template <int VEC, int MODE>
__global__ __launch_bounds__(256) void segment_csr_kernel(
    const float *__restrict__ src, int C, const i64 *__restrict__ ptr, i64 nseg,
    float *__restrict__ out)
{
    typedef typename Vec<VEC>::T V;
    const unsigned cv = (unsigned)(C / VEC);             // row length in vectors
    const i64 total = nseg * cv;
    const V *s = (const V *)src;
    V *o = (V *)out;
    const i64 e0 = (i64)blockIdx.x * (256 * SEG_ILP) + threadIdx.x;
    // one 64-bit division per block, 32-bit ones per pair
    const i64 seg0 = ((i64)blockIdx.x * (256 * SEG_ILP)) / cv;
    const unsigned rem0 = (unsigned)((i64)blockIdx.x * (256 * SEG_ILP) - seg0 * cv) + threadIdx.x;
    i64 a[SEG_ILP], b[SEG_ILP];
    unsigned c[SEG_ILP];
    V acc[SEG_ILP];
#pragma unroll
    for (int k = 0; k < SEG_ILP; ++k) {
        const i64 e = e0 + k * 256;
        a[k] = b[k] = 0;
        c[k] = 0;
        if (e < total) {
            const unsigned off = rem0 + k * 256, ds = off / cv;
            const i64 seg = seg0 + ds;
            c[k] = off - ds * cv;
            a[k] = ptr[seg];
            b[k] = ptr[seg + 1];
        }
    }
    // first rows of all pairs together, then the remaining rows pair by pair
#pragma unroll
    for (int k = 0; k < SEG_ILP; ++k) {
        zero_v(acc[k]);
        if (b[k] > a[k]) acc[k] = s[(size_t)a[k] * cv + c[k]];
    }
#pragma unroll
    for (int k = 0; k < SEG_ILP; ++k) {
        // four rows at a time, loaded together (rows past the end predicated off),
        // folded in row order
        for (i64 r = a[k] + 1; r < b[k]; r += 4) {
            const V *q = s + (size_t)r * cv + c[k];
            const bool v1 = r + 1 < b[k], v2 = r + 2 < b[k], v3 = r + 3 < b[k];
            V x0 = q[0], x1, x2, x3;
            zero_v(x1); zero_v(x2); zero_v(x3);
            if (v1) x1 = q[(size_t)cv];
            if (v2) x2 = q[2 * (size_t)cv];
            if (v3) x3 = q[3 * (size_t)cv];
            fold_v<MODE>(acc[k], x0);
            if (v1) fold_v<MODE>(acc[k], x1);
            if (v2) fold_v<MODE>(acc[k], x2);
            if (v3) fold_v<MODE>(acc[k], x3);
        }
        if (MODE == MVF_REDUCE_MEAN && b[k] > a[k]) scale_v(acc[k], (float)(b[k] - a[k]));
        const i64 e = e0 + k * 256;
        if (e < total) o[e] = acc[k];
    }
}

// max / min over long segments (many rows per segment, e.g. the pixels of a
// splat): the serial walk above keeps one thread on a segment and is bound by
// its chain of loads.  Here a wave takes a segment: lane = (row group, channel)
// with 64 >> cp_log row groups of 1 << cp_log >= min(C, 64) channels,
// SEG_LONG_LOADS steps of rows in flight, and a final combine across the row
// groups.  The result is bit-identical to segment_csr_kernel's in-order fold
// (acc = first row; x > acc replaces): ties (+0.0 / -0.0 compare equal) go to the
// first row, a NaN sticks when it is the segment's first row and is passed over
// anywhere else.
//
// The kernel is bound by vector issue (a million waves of a few dozen rows each),
// so the common case is kept to five instructions per load step: v_max_f32 /
// v_min_f32 on values preset to -inf / +inf, which gives the fold's result
// whenever no row holds a NaN and the result is not a zero (the only cases in
// which the order of the rows shows).  A wave that saw a NaN, or whose result is
// +-0 in some channel, repeats the segment with the exact procedure
// (seg_long_exact: row index of every candidate kept, first row wins).
#ifndef MVF_SEG_LONG_LOADS
#define MVF_SEG_LONG_LOADS 16
#endif
constexpr int SEG_LONG_LOADS = MVF_SEG_LONG_LOADS;       // rows a lane has in flight

template <int MODE>
__device__ __forceinline__ float seg_long_exact(
    const float *q, unsigned n, bool live, int ro, int rp, int cp, int C)
{
    auto better = [](float x, float y) { return MODE == MVF_REDUCE_MAX ? x > y : x < y; };
    float acc = 0.f;
    unsigned at = 0xffffffffu;                        // row (relative to the segment) the value comes from
    for (unsigned r = (unsigned)ro; r < n && live; r += (unsigned)(SEG_LONG_LOADS * rp)) {
        float x[SEG_LONG_LOADS];
        bool ok[SEG_LONG_LOADS];
#pragma unroll
        for (int k = 0; k < SEG_LONG_LOADS; ++k) {
            ok[k] = r + (unsigned)(k * rp) < n;
            x[k] = ok[k] ? q[(size_t)(r + (unsigned)(k * rp)) * C] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < SEG_LONG_LOADS; ++k) {
            const unsigned rk = r + (unsigned)(k * rp);
            // a lane starts from its first number (or from row 0, whatever it holds)
            const bool start = at == 0xffffffffu && (x[k] == x[k] || rk == 0u);
            if (ok[k] && (start || better(x[k], acc))) { acc = x[k]; at = rk; }
        }
    }
    // combine the row groups (lane bits cp_log .. 5)
    for (int d = 32; d >= cp; d >>= 1) {
        const float ov = __shfl_xor(acc, d, 64);
        const unsigned oat = __shfl_xor(at, d, 64);
        // (only row 0 can have left a NaN: it wins over everything)
        const bool take = oat != 0xffffffffu && acc == acc &&
            (at == 0xffffffffu || ov != ov || better(ov, acc) || (!better(acc, ov) && oat < at));
        if (take) { acc = ov; at = oat; }
    }
    return at == 0xffffffffu ? 0.f : acc;
}

// v_max_f32 / v_min_f32 as written (the builtins add an instruction per operand to quiet
// signalling NaNs; a wave that meets any NaN takes the exact procedure anyway)
template <int MODE> __device__ __forceinline__ float seg_fold_raw(float a, float x)
{
    float r;
    if (MODE == MVF_REDUCE_MAX) asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(x));
    else asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(x));
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void segment_long_kernel(
    const float *__restrict__ src, int C, int cp_log, const i64 *__restrict__ ptr, i64 nseg,
    float *__restrict__ out)
{
    // the wave's segment as a scalar: its bounds come through scalar loads
    const i64 seg = (i64)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (seg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const int cp = 1 << cp_log, rp = 64 >> cp_log;
    const int c = lane & (cp - 1), ro = lane >> cp_log;
    const i64 a = ptr[seg], b = ptr[seg + 1];
    const unsigned n = (unsigned)(b - a);
    const float worst = MODE == MVF_REDUCE_MAX ? -INFINITY : INFINITY;
    // Addresses inside the segment are 32-bit byte offsets from its first row, a scalar base:
    // two vector instructions per load (add, clamp) and the load takes the base from scalar
    // registers.  A segment of 2^32 bytes or more takes the exact procedure right away.
    const char *base = (const char *)(src + (size_t)a * C);
    const bool huge = ((unsigned long long)n * (unsigned)C) >> 30;
    const unsigned step = (unsigned)(rp * C) * 4u;    // bytes between two load steps of a lane
    for (int c0 = 0; c0 < C; c0 += cp) {
        const int ch = c0 + c;
        const bool live = ch < C;
        const float *q = src + (size_t)a * C + ch;
        float acc = worst;
        float chk = 0.f;                              // turns NaN when a row holds a NaN (or an infinity)
        // Steps past the end read the segment's last row again (same channel), which changes
        // neither a maximum nor a minimum: no lane mask inside the batch.
        const unsigned last = ((n - 1u) * (unsigned)C + (unsigned)ch) * 4u;
        unsigned off = ((unsigned)ro * (unsigned)C + (unsigned)ch) * 4u;
        for (unsigned r = (unsigned)ro; r < n && live && !huge; r += (unsigned)(SEG_LONG_LOADS * rp)) {
            float x[SEG_LONG_LOADS];
#pragma unroll
            for (int k = 0; k < SEG_LONG_LOADS; ++k) {
                const unsigned o = off + (unsigned)k * step;
                x[k] = *(const float *)(base + (o < last ? o : last));
            }
#pragma unroll
            for (int k = 0; k < SEG_LONG_LOADS; ++k) {
                chk = __fmaf_rn(x[k], 0.f, chk);
                acc = seg_fold_raw<MODE>(acc, x[k]);
            }
            off += (unsigned)SEG_LONG_LOADS * step;
        }
        for (int d = 32; d >= cp; d >>= 1) acc = seg_fold_raw<MODE>(acc, __shfl_xor(acc, d, 64));
        // the order of the rows shows in the result: a NaN somewhere, or a zero that wins
        // (+0.0 and -0.0 tie and the first row keeps its sign)
        if (huge || __ballot(live && (chk != chk || acc == 0.f)) != 0)
            acc = seg_long_exact<MODE>(q, n, live, ro, rp, cp, C);
        if (ro == 0 && live) out[(size_t)seg * C + ch] = n == 0 ? 0.f : acc;
    }
}

// Two pooling levels in one pass:
//   out_x = segment_csr(segment_csr(src, inner_ptr, INNER), outer_ptr, reduce_x)
// for up to two outer reductions, without the [n_inner][C] intermediate and
// with src read once.  Each inner segment is folded in row order and fed to
// running sum / min / max accumulators in inner-segment order: the same
// operations in the same order as the two separate passes, so the results are
// identical.
template <int MODE> __device__ __forceinline__ void fold_first_v(float &a, const float &x, bool first) { a = first ? x : fold<MODE>(a, x); }
template <int MODE> __device__ __forceinline__ void fold_first_v(float4 &a, const float4 &x, bool first)
{
    a.x = first ? x.x : fold<MODE>(a.x, x.x); a.y = first ? x.y : fold<MODE>(a.y, x.y);
    a.z = first ? x.z : fold<MODE>(a.z, x.z); a.w = first ? x.w : fold<MODE>(a.w, x.w);
}

template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T pick_outer(
    int reduce, const typename Vec<VEC>::T &sum, const typename Vec<VEC>::T &mn,
    const typename Vec<VEC>::T &mx, i64 count)
{
    typename Vec<VEC>::T r;
    zero_v(r);
    if (count > 0) {
        r = reduce == MVF_REDUCE_MAX ? mx : reduce == MVF_REDUCE_MIN ? mn : sum;
        if (reduce == MVF_REDUCE_MEAN) scale_v(r, (float)count);
    }
    return r;
}

// The output is walked like segment_csr's: a thread owns one (outer segment,
// column vector) pair, consecutive threads own consecutive vectors.  Inner
// segments are taken TL_VU at a time: their pointers and their first rows are
// loaded together (independent loads: the chain outer pointer -> inner pointer
// -> row is latency-bound otherwise), the folds happen in inner-segment order
// afterwards.
constexpr int TL_VU = 4;

template <int VEC, int INNER>
__global__ __launch_bounds__(256) void two_level_pool_kernel(
    const float *__restrict__ src, int C, const i64 *__restrict__ inner_ptr,
    const i64 *__restrict__ outer_ptr, i64 n_outer, int reduce_a, float *__restrict__ out_a,
    int reduce_b, float *__restrict__ out_b)
{
    typedef typename Vec<VEC>::T V;
    const unsigned cv = (unsigned)(C / VEC);
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_outer * cv) return;
    // (a 32-bit division where the index allows it: the 64-bit one is a hundred instructions)
    const i64 seg = n_outer * cv <= 0xffffffffll ? (i64)((unsigned)e / cv) : e / cv;
    const V *s = (const V *)src + (unsigned)(e - seg * cv);
    const i64 va = outer_ptr[seg], vb = outer_ptr[seg + 1];
    V sum, mn, mx;
    zero_v(sum); zero_v(mn); zero_v(mx);
    for (i64 r = va; r < vb; r += TL_VU) {
        i64 q[TL_VU + 1];
#pragma unroll
        for (int u = 0; u <= TL_VU; ++u) q[u] = inner_ptr[r + u < vb ? r + u : vb];
        V acc[TL_VU];
#pragma unroll
        for (int u = 0; u < TL_VU; ++u) {
            zero_v(acc[u]);
            if (r + u < vb && q[u + 1] > q[u]) acc[u] = s[(size_t)q[u] * cv];
        }
#pragma unroll
        for (int u = 0; u < TL_VU; ++u) {
            if (r + u >= vb) break;
            for (i64 p = q[u] + 1; p < q[u + 1]; ++p) fold_v<INNER>(acc[u], s[(size_t)p * cv]);
            if (INNER == MVF_REDUCE_MEAN && q[u + 1] > q[u]) scale_v(acc[u], (float)(q[u + 1] - q[u]));
            const bool first = r + u == va;
            fold_first_v<MVF_REDUCE_SUM>(sum, acc[u], first);
            fold_first_v<MVF_REDUCE_MIN>(mn, acc[u], first);
            fold_first_v<MVF_REDUCE_MAX>(mx, acc[u], first);
        }
    }
    ((V *)out_a)[e] = pick_outer<VEC>(reduce_a, sum, mn, mx, vb - va);
    if (out_b) ((V *)out_b)[e] = pick_outer<VEC>(reduce_b, sum, mn, mx, vb - va);
}

template <int VEC>
__global__ __launch_bounds__(256) void gather_csr_kernel(
    const float *__restrict__ src, int C, const i64 *__restrict__ ptr, i64 nseg,
    int lpr_log, float *__restrict__ out)
{
    typedef typename Vec<VEC>::T V;
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 seg = gtid >> lpr_log;
    if (seg >= nseg) return;
    const int lig = (int)(gtid & ((1 << lpr_log) - 1));
    const i64 a = ptr[seg], b = ptr[seg + 1];
    const int cv = C / VEC;
    const V *s = (const V *)src;
    V *o = (V *)out;
    for (int c = lig; c < cv; c += (1 << lpr_log)) {
        const V val = s[(size_t)seg * cv + c];
        for (i64 r = a; r < b; ++r) o[(size_t)r * cv + c] = val;
    }
}

// One thread per (segment, group column): segments are short (<= number of
// views), the three sweeps stay in L1/L2.
__global__ __launch_bounds__(256) void segment_softmax_kernel(
    const float *__restrict__ src, int G, const i64 *__restrict__ ptr, i64 nseg,
    float eps, int scaling, float *__restrict__ out)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 seg = t / G;
    if (seg >= nseg) return;
    const int g = (int)(t - seg * G);
    const i64 a = ptr[seg], b = ptr[seg + 1];
    if (b <= a) return;
    float mx = src[(size_t)a * G + g];
    for (i64 r = a + 1; r < b; ++r) { const float v = src[(size_t)r * G + g]; mx = v > mx ? v : mx; }
    const float sq = sqrtf((float)(b - a));
    float sum = 0.f;
    for (i64 r = a; r < b; ++r) {
        float c = __fsub_rn(src[(size_t)r * G + g], mx);
        if (scaling) c = __fdiv_rn(c, sq);
        const float e = expf(c);
        out[(size_t)r * G + g] = e;
        sum = (r == a) ? e : __fadd_rn(sum, e);
    }
    sum = __fadd_rn(sum, eps);
    for (i64 r = a; r < b; ++r) out[(size_t)r * G + g] = __fdiv_rn(out[(size_t)r * G + g], sum);
}

// Feature-map gather fused with the atomic (pixel -> view) pool:
// out[r][c] = reduce_{p in [vptr[r], vptr[r+1])} x[images[r]][c][py[p]][px[p]]
// for x of shape [B][C][H][W] -- the reference materialises x[idx] as [P][C]
// first (core/multimodal/image.py:1272-1297, 1881-1895) and pools it afterwards
// (pooling.py:63).  A block owns 64 view rows: lane = row, so that the gathers
// of one channel plane hit neighbouring pixels of neighbouring points, four
// channels in flight per row; the [64][64-channel] tile goes through LDS so that
// the output rows are written contiguously.
template <int MODE>
__global__ __launch_bounds__(256) void gather_pool_kernel(
    const float *__restrict__ x, int C, int H, int W, const i64 *__restrict__ images,
    const i64 *__restrict__ vptr, const int16_t *__restrict__ pixels, i64 rows,
    float *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int row = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const i64 r = (i64)blockIdx.x * 64 + row;
    i64 a = 0, b = 0;
    const float *plane = x;
    if (r < rows) {
        a = vptr[r]; b = vptr[r + 1];
        plane = x + (size_t)images[r] * C * H * W;
    }
    const size_t hw = (size_t)H * W;
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int c = c0 + cq; c < C && c < c0 + 64; c += 4) {
            const float *pc = plane + (size_t)c * hw;
            float acc = 0.f;
            for (i64 p = a; p < b; ++p) {
                const float v = pc[(size_t)pixels[2 * p + 1] * W + pixels[2 * p]];
                acc = p == a ? v : fold<MODE>(acc, v);
            }
            if (MODE == MVF_REDUCE_MEAN && b > a) acc = __fdiv_rn(acc, (float)(b - a));
            tile[row][c - c0] = acc;
        }
        __syncthreads();
        const int cw = C - c0 < 64 ? C - c0 : 64;
        for (int e = threadIdx.x; e < 64 * cw; e += 256) {
            const int rr = e / cw, cc = e - rr * cw;
            const i64 ro = (i64)blockIdx.x * 64 + rr;
            if (ro < rows) out[(size_t)ro * C + c0 + cc] = tile[rr][cc];
        }
        __syncthreads();
    }
}

// Feature-map gather: out[p][c] = x[batch[p]][c][py[p]][px[p]] for x of shape
// [B][C][H][W] -- X[mappings.feature_map_indexing] of the reference
// (core/multimodal/image.py:1295, 1881-1895).  Lane = mapped pixel, four
// channels in flight per lane; the [64][64-channel] tile goes through LDS so
// that the output rows are written contiguously.
__global__ __launch_bounds__(256) void gather_features_kernel(
    const float *__restrict__ x, int C, int H, int W, const i64 *__restrict__ batch,
    const int16_t *__restrict__ pixels, i64 rows, float *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int row = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const i64 r = (i64)blockIdx.x * 64 + row;
    const size_t hw = (size_t)H * W;
    const float *px = x;
    if (r < rows) px = x + (size_t)batch[r] * C * hw + (size_t)pixels[2 * r + 1] * W + pixels[2 * r];
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int c = c0 + cq; c < C && c < c0 + 64; c += 4)
            tile[row][c - c0] = r < rows ? px[(size_t)c * hw] : 0.f;
        __syncthreads();
        const int cw = C - c0 < 64 ? C - c0 : 64;
        for (int e = threadIdx.x; e < 64 * cw; e += 256) {
            const int rr = e / cw, cc = e - rr * cw;
            const i64 ro = (i64)blockIdx.x * 64 + rr;
            if (ro < rows) out[(size_t)ro * C + c0 + cc] = tile[rr][cc];
        }
        __syncthreads();
    }
}

// Index of padded row / column i (0 .. n + 1) in the unpadded map, or -1 for a
// zero: ZeroPad2d(1), ReplicationPad2d(1), ReflectionPad2d(1).
enum { PAD_BORDER = 0, PAD_ZEROS = 1, PAD_REFLECTION = 2 };

__device__ __forceinline__ int unpad(int i, int n, int mode)
{
    const int j = i - 1;
    if (j >= 0 && j < n) return j;
    if (mode == PAD_ZEROS) return -1;
    if (mode == PAD_BORDER) return j < 0 ? 0 : n - 1;
    return j < 0 ? 1 : n - 2;
}

// sparse_interpolation (core/multimodal/image.py:105-170): bilinear sampling
// of x [B][C][H][W] at coords [n][2] = (row, column) in [0, 1], one feature map
// per query (batch), on the map padded by one pixel.  Weights and sums are
// float32 operations in the reference's order.
__global__ __launch_bounds__(256) void sparse_interp_kernel(
    const float *__restrict__ x, int C, int H, int W, const float *__restrict__ coords,
    const i64 *__restrict__ batch, i64 rows, int mode, float *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int row = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const i64 r = (i64)blockIdx.x * 64 + row;
    const size_t hw = (size_t)H * W;
    const float *plane = x;
    float w_tl = 0.f, w_tr = 0.f, w_bl = 0.f, w_br = 0.f;
    i64 o_tl = -1, o_tr = -1, o_bl = -1, o_br = -1;
    if (r < rows) {
        plane = x + (size_t)batch[r] * C * hw;
        const float p0 = __fadd_rn(__fmul_rn(coords[2 * r], (float)H), 0.5f);
        const float p1 = __fadd_rn(__fmul_rn(coords[2 * r + 1], (float)W), 0.5f);
        const float top = floorf(p0), bottom = floorf(__fadd_rn(p0, 1.0f));
        const float left = floorf(p1), right = floorf(__fadd_rn(p1, 1.0f));
        w_tl = fabsf(__fmul_rn(__fsub_rn(p0, bottom), __fsub_rn(p1, right)));
        w_tr = fabsf(__fmul_rn(__fsub_rn(p0, bottom), __fsub_rn(p1, left)));
        w_bl = fabsf(__fmul_rn(__fsub_rn(p0, top), __fsub_rn(p1, right)));
        w_br = fabsf(__fmul_rn(__fsub_rn(p0, top), __fsub_rn(p1, left)));
        const int t = unpad((int)top, H, mode), b = unpad((int)bottom, H, mode);
        const int l = unpad((int)left, W, mode), rt = unpad((int)right, W, mode);
        if (t >= 0 && l >= 0) o_tl = (i64)t * W + l;
        if (t >= 0 && rt >= 0) o_tr = (i64)t * W + rt;
        if (b >= 0 && l >= 0) o_bl = (i64)b * W + l;
        if (b >= 0 && rt >= 0) o_br = (i64)b * W + rt;
    }
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int c = c0 + cq; c < C && c < c0 + 64; c += 4) {
            const float *pc = plane + (size_t)c * hw;
            const float v_tl = o_tl >= 0 ? pc[o_tl] : 0.f, v_tr = o_tr >= 0 ? pc[o_tr] : 0.f;
            const float v_bl = o_bl >= 0 ? pc[o_bl] : 0.f, v_br = o_br >= 0 ? pc[o_br] : 0.f;
            float acc = __fadd_rn(__fmul_rn(w_tl, v_tl), __fmul_rn(w_tr, v_tr));
            acc = __fadd_rn(acc, __fmul_rn(w_bl, v_bl));
            acc = __fadd_rn(acc, __fmul_rn(w_br, v_br));
            tile[row][c - c0] = acc;
        }
        __syncthreads();
        const int cw = C - c0 < 64 ? C - c0 : 64;
        for (int e = threadIdx.x; e < 64 * cw; e += 256) {
            const int rr = e / cw, cc = e - rr * cw;
            const i64 ro = (i64)blockIdx.x * 64 + rr;
            if (ro < rows) out[(size_t)ro * C + c0 + cc] = tile[rr][cc];
        }
        __syncthreads();
    }
}

// Gradient of x[feature_map_indexing] with respect to x: row r's gradient is added
// to the pixel it was read from (several rows may share a pixel, hence atomics;
// the sum of a pixel's rows is order dependent in the last bits, as the
// accumulating index_put of torch's own gather gradient is).  Lanes walk the
// rows, so the channel planes are written with the access pattern the forward
// read them with.
__global__ __launch_bounds__(256) void scatter_features_kernel(
    const float *__restrict__ grad, int C, int H, int W, const i64 *__restrict__ batch,
    const int16_t *__restrict__ pixels, i64 rows, float *__restrict__ grad_x)
{
    const int row = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const i64 r = (i64)blockIdx.x * 64 + row;
    if (r >= rows) return;
    const size_t hw = (size_t)H * W;
    float *px = grad_x + (size_t)batch[r] * C * hw + (size_t)pixels[2 * r + 1] * W + pixels[2 * r];
    for (int c = cq; c < C; c += 4) unsafeAtomicAdd(px + (size_t)c * hw, grad[(size_t)r * C + c]);
}

// Gradient of sparse_interpolation with respect to the feature maps: the four
// weights of the forward, times the row's gradient, added to the four pixels.
__global__ __launch_bounds__(256) void sparse_interp_backward_kernel(
    const float *__restrict__ grad, int C, int H, int W, const float *__restrict__ coords,
    const i64 *__restrict__ batch, i64 rows, int mode, float *__restrict__ grad_x)
{
    const int row = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const i64 r = (i64)blockIdx.x * 64 + row;
    if (r >= rows) return;
    const size_t hw = (size_t)H * W;
    float *plane = grad_x + (size_t)batch[r] * C * hw;
    const float p0 = __fadd_rn(__fmul_rn(coords[2 * r], (float)H), 0.5f);
    const float p1 = __fadd_rn(__fmul_rn(coords[2 * r + 1], (float)W), 0.5f);
    const float top = floorf(p0), bottom = floorf(__fadd_rn(p0, 1.0f));
    const float left = floorf(p1), right = floorf(__fadd_rn(p1, 1.0f));
    const float w_tl = fabsf(__fmul_rn(__fsub_rn(p0, bottom), __fsub_rn(p1, right)));
    const float w_tr = fabsf(__fmul_rn(__fsub_rn(p0, bottom), __fsub_rn(p1, left)));
    const float w_bl = fabsf(__fmul_rn(__fsub_rn(p0, top), __fsub_rn(p1, right)));
    const float w_br = fabsf(__fmul_rn(__fsub_rn(p0, top), __fsub_rn(p1, left)));
    const int t = unpad((int)top, H, mode), b = unpad((int)bottom, H, mode);
    const int l = unpad((int)left, W, mode), rt = unpad((int)right, W, mode);
    for (int c = cq; c < C; c += 4) {
        const float g = grad[(size_t)r * C + c];
        float *pc = plane + (size_t)c * hw;
        if (t >= 0 && l >= 0) unsafeAtomicAdd(pc + (size_t)t * W + l, w_tl * g);
        if (t >= 0 && rt >= 0) unsafeAtomicAdd(pc + (size_t)t * W + rt, w_tr * g);
        if (b >= 0 && l >= 0) unsafeAtomicAdd(pc + (size_t)b * W + l, w_bl * g);
        if (b >= 0 && rt >= 0) unsafeAtomicAdd(pc + (size_t)b * W + rt, w_br * g);
    }
}

// Backward of segment_csr (the gradient torch_scatter attaches to it): one
// thread per (segment, channel).  sum: every row of the segment receives the
// segment's gradient; mean: divided by the row count; max / min: the FIRST row
// that holds the reduced value receives it (the forward keeps the first of
// equal values), the other rows zero.
template <int MODE>
__global__ __launch_bounds__(256) void segment_csr_backward_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ src, const float *__restrict__ out,
    int C, const i64 *__restrict__ ptr, i64 nseg, float *__restrict__ grad_src)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= nseg * C) return;
    const i64 sg = e / C;
    const int c = (int)(e - sg * C);
    const i64 a = ptr[sg], b = ptr[sg + 1];
    if (b <= a) return;
    float g = grad_out[e];
    if (MODE == MVF_REDUCE_SUM || MODE == MVF_REDUCE_MEAN) {
        if (MODE == MVF_REDUCE_MEAN) g = __fdiv_rn(g, (float)(b - a));
        for (i64 r = a; r < b; ++r) grad_src[(size_t)r * C + c] = g;
    } else {
        const float best = out[e];
        bool given = false;
        for (i64 r = a; r < b; ++r) {
            const bool hit = !given && src[(size_t)r * C + c] == best;
            grad_src[(size_t)r * C + c] = hit ? g : 0.f;
            given = given || hit;
        }
    }
}

// Backward of segment_softmax_csr: with y = softmax of x / s inside the segment
// (s = sqrt(rows) when scaling), grad_x = y * (g - sum_rows(g * y)) / s.  The
// reference's eps in the denominator and its max-centring change the gradient
// by O(eps) and are not differentiated through.  One thread per (segment,
// group), rows folded in order.
__global__ __launch_bounds__(256) void segment_softmax_backward_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ y, int G,
    const i64 *__restrict__ ptr, i64 nseg, int scaling, float *__restrict__ grad_src)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= nseg * G) return;
    const i64 sg = e / G;
    const int g = (int)(e - sg * G);
    const i64 a = ptr[sg], b = ptr[sg + 1];
    if (b <= a) return;
    float dot = 0.f;
    for (i64 r = a; r < b; ++r) dot = __fmaf_rn(grad_out[(size_t)r * G + g], y[(size_t)r * G + g], dot);
    const float inv = scaling ? __fdiv_rn(1.0f, sqrtf((float)(b - a))) : 1.0f;
    for (i64 r = a; r < b; ++r) {
        const size_t k = (size_t)r * G + g;
        grad_src[k] = __fmul_rn(__fmul_rn(y[k], __fsub_rn(grad_out[k], dot)), inv);
    }
}

static int pick_layout(const void *p0, const void *p1, int C, int &vec, int &lpr_log)
{
    vec = (C % 4 == 0 && ((uintptr_t)p0 % 16 == 0) && ((uintptr_t)p1 % 16 == 0)) ? 4 : 1;
    int lanes = C / vec, l = 0;
    while ((1 << l) < lanes && l < 6) ++l;
    lpr_log = l;
    return 0;
}

// Attention-weighted sum of the rows of every segment (the pooling step of the attentive
// pools, pooling.py:289-296, 519-526, with a graph): out[s][c] = sum over the rows r of s of
// x[r][c] * att[r][group of c], G groups of C / G adjacent channels, rows folded in order.
// A thread owns 4 channels of one segment.
__global__ __launch_bounds__(256) void weighted_segment_sum_kernel(
    const float *__restrict__ x, const float *__restrict__ att, int C, int G,
    const i64 *__restrict__ ptr, i64 nseg, float *__restrict__ out)
{
    const unsigned cv = (unsigned)(C / 4);
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= nseg * cv) return;
    const i64 seg = e / cv;
    const unsigned c4 = (unsigned)(e - seg * cv);
    const unsigned g = c4 * 4u / (unsigned)(C / G);
    const i64 a = ptr[seg], b = ptr[seg + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *xs = (const float4 *)x;
    for (i64 r = a; r < b; r += 2) {
        const bool two = r + 1 < b;
        const float4 v0 = xs[(size_t)r * cv + c4];
        const float w0 = att[(size_t)r * G + g];
        float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        float w1 = 0.f;
        if (two) {
            v1 = xs[(size_t)(r + 1) * cv + c4];
            w1 = att[(size_t)(r + 1) * G + g];
        }
        acc.x = __fadd_rn(acc.x, __fmul_rn(v0.x, w0)); acc.y = __fadd_rn(acc.y, __fmul_rn(v0.y, w0));
        acc.z = __fadd_rn(acc.z, __fmul_rn(v0.z, w0)); acc.w = __fadd_rn(acc.w, __fmul_rn(v0.w, w0));
        if (two) {
            acc.x = __fadd_rn(acc.x, __fmul_rn(v1.x, w1)); acc.y = __fadd_rn(acc.y, __fmul_rn(v1.y, w1));
            acc.z = __fadd_rn(acc.z, __fmul_rn(v1.z, w1)); acc.w = __fadd_rn(acc.w, __fmul_rn(v1.w, w1));
        }
    }
    ((float4 *)out)[e] = acc;
}

// Its gradients: grad_x[r][c] = grad_out[seg r][c] * att[r][g(c)] and grad_att[r][g] = sum over
// the channels of g of grad_out[seg r][c] * x[r][c] (zero for a row whose row2seg entry is no
// segment index).  A thread owns 4 channels of one row; the
// C / (4 G) threads of a group are adjacent lanes of one wave (a power of two) and add their
// partial sums by xor shuffles (a fixed tree).
__global__ __launch_bounds__(256) void weighted_segment_sum_backward_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ x, const float *__restrict__ att,
    int C, int G, const int *__restrict__ row2seg, i64 rows, i64 nseg,
    float *__restrict__ grad_x, float *__restrict__ grad_att)
{
    const unsigned cv = (unsigned)(C / 4);
    const unsigned lg = cv / (unsigned)G;               // lanes per group
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool live = e < rows * cv;                    // (whole rows per wave: cv divides 64)
    const i64 r = live ? e / cv : 0;
    const unsigned c4 = live ? (unsigned)(e - r * cv) : 0u;
    const unsigned g = c4 / lg;
    float dot = 0.f;
    if (live) {
        const i64 seg = row2seg[r];
        // (a row of no segment: zero gradients)
        const float4 go = seg >= 0 && seg < nseg ? ((const float4 *)grad_out)[(size_t)seg * cv + c4]
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 xv = ((const float4 *)x)[(size_t)r * cv + c4];
        const float w = att[(size_t)r * G + g];
        if (grad_x)
            ((float4 *)grad_x)[e] = make_float4(__fmul_rn(go.x, w), __fmul_rn(go.y, w), __fmul_rn(go.z, w), __fmul_rn(go.w, w));
        dot = __fadd_rn(__fadd_rn(__fmul_rn(go.x, xv.x), __fmul_rn(go.y, xv.y)),
                        __fadd_rn(__fmul_rn(go.z, xv.z), __fmul_rn(go.w, xv.w)));
    }
    for (unsigned d = 1; d < lg; d <<= 1) dot = __fadd_rn(dot, __shfl_xor(dot, (int)d, 64));
    if (live && grad_att && (c4 & (lg - 1)) == 0) grad_att[(size_t)r * G + g] = dot;
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_segment_csr(
    const float *src, int64_t rows, int32_t C, const int64_t *ptr, int64_t nseg,
    int32_t reduce, float *out, void *stream)
{
    MVF_REQUIRE(C >= 1 && nseg >= 0 && rows >= 0, "segment_csr: bad shape C=%d nseg=%lld", C, (long long)nseg);
    MVF_REQUIRE(reduce >= 0 && reduce <= 3, "segment_csr: unknown reduce %d", reduce);
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(ptr && out && (src || rows == 0), "segment_csr: null pointer");
    int vec, lpr_log;
    pick_layout(src, out, C, vec, lpr_log);
    hipStream_t st = (hipStream_t)stream;
    if ((reduce == MVF_REDUCE_MAX || reduce == MVF_REDUCE_MIN) && rows >= 16 * nseg) {
        // long segments on average: a wave per segment
        int cp_log = 0;
        while ((1 << cp_log) < C && cp_log < 6) ++cp_log;
        if (reduce == MVF_REDUCE_MAX)
            hipLaunchKernelGGL((segment_long_kernel<MVF_REDUCE_MAX>), dim3(cdiv(nseg, 4)), dim3(256), 0, st, src, C,
                               cp_log, (const i64 *)ptr, (i64)nseg, out);
        else
            hipLaunchKernelGGL((segment_long_kernel<MVF_REDUCE_MIN>), dim3(cdiv(nseg, 4)), dim3(256), 0, st, src, C,
                               cp_log, (const i64 *)ptr, (i64)nseg, out);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    dim3 grid(cdiv(nseg * (C / vec), 256 * SEG_ILP)), block(256);
#define MVF_SEG(V_, M_) \
    hipLaunchKernelGGL((segment_csr_kernel<V_, M_>), grid, block, 0, st, src, C, (const i64 *)ptr, (i64)nseg, out)
#define MVF_SEG_MODE(V_)                                        \
    switch (reduce) {                                           \
    case MVF_REDUCE_SUM: MVF_SEG(V_, MVF_REDUCE_SUM); break;    \
    case MVF_REDUCE_MEAN: MVF_SEG(V_, MVF_REDUCE_MEAN); break;  \
    case MVF_REDUCE_MIN: MVF_SEG(V_, MVF_REDUCE_MIN); break;    \
    default: MVF_SEG(V_, MVF_REDUCE_MAX); break;                \
    }
    if (vec == 4) { MVF_SEG_MODE(4) } else { MVF_SEG_MODE(1) }
#undef MVF_SEG_MODE
#undef MVF_SEG
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_segment_csr_backward(
    const float *grad_out, const float *src, const float *out, int64_t rows, int32_t C,
    const int64_t *ptr, int64_t nseg, int32_t reduce, float *grad_src, void *stream)
{
    MVF_REQUIRE(C >= 1 && nseg >= 0 && rows >= 0, "segment_csr_backward: bad shape");
    MVF_REQUIRE(reduce >= 0 && reduce <= 3, "segment_csr_backward: unknown reduce %d", reduce);
    if (nseg == 0 || rows == 0) return MVF_OK;
    MVF_REQUIRE(grad_out && ptr && grad_src, "segment_csr_backward: null pointer");
    const bool sel = reduce == MVF_REDUCE_MIN || reduce == MVF_REDUCE_MAX;
    MVF_REQUIRE(!sel || (src && out), "segment_csr_backward: max / min need src and out");
    dim3 grid(cdiv(nseg * C, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MVF_SB(M_) hipLaunchKernelGGL((segment_csr_backward_kernel<M_>), grid, block, 0, st, grad_out, src, \
                                      out, C, (const i64 *)ptr, (i64)nseg, grad_src)
    switch (reduce) {
    case MVF_REDUCE_SUM: MVF_SB(MVF_REDUCE_SUM); break;
    case MVF_REDUCE_MEAN: MVF_SB(MVF_REDUCE_MEAN); break;
    case MVF_REDUCE_MIN: MVF_SB(MVF_REDUCE_MIN); break;
    default: MVF_SB(MVF_REDUCE_MAX); break;
    }
#undef MVF_SB
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

static bool weighted_sum_shape_ok(int C, int G)
{
    if (C < 4 || G < 1 || C % 4 != 0 || C % G != 0 || (C / G) % 4 != 0) return false;
    const int cv = C / 4, lg = cv / G;
    return cv <= 64 && (cv & (cv - 1)) == 0 && (lg & (lg - 1)) == 0;
}

extern "C" int mvf_weighted_segment_sum(
    const float *x, const float *att, int32_t C, int32_t G, const int64_t *ptr, int64_t nseg,
    float *out, void *stream)
{
    MVF_REQUIRE(weighted_sum_shape_ok(C, G),
                "weighted_segment_sum: C / 4 and C / (4 G) must be powers of two, C <= 256 (got C=%d G=%d)", C, G);
    MVF_REQUIRE(nseg >= 0 && out && ptr, "weighted_segment_sum: null argument");
    MVF_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "weighted_segment_sum: x and out must be 16-byte aligned");
    if (nseg == 0) return MVF_OK;
    hipLaunchKernelGGL(weighted_segment_sum_kernel, dim3(cdiv(nseg * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       x, att, C, G, (const i64 *)ptr, (i64)nseg, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_weighted_segment_sum_backward(
    const float *grad_out, const float *x, const float *att, int32_t C, int32_t G,
    const int32_t *row2seg, int64_t rows, int64_t nseg, float *grad_x, float *grad_att, void *stream)
{
    MVF_REQUIRE(weighted_sum_shape_ok(C, G),
                "weighted_segment_sum_backward: C / 4 and C / (4 G) must be powers of two, C <= 256 (got C=%d G=%d)", C, G);
    MVF_REQUIRE(rows >= 0 && grad_out && row2seg, "weighted_segment_sum_backward: null argument");
    MVF_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)grad_out % 16 == 0 && (uintptr_t)grad_x % 16 == 0,
                "weighted_segment_sum_backward: x, grad_out and grad_x must be 16-byte aligned");
    if (rows == 0) return MVF_OK;
    hipLaunchKernelGGL(weighted_segment_sum_backward_kernel, dim3(cdiv(rows * (C / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, x, att, C, G, row2seg, (i64)rows, (i64)nseg, grad_x, grad_att);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_gather_csr(
    const float *src, int32_t C, const int64_t *ptr, int64_t nseg, int64_t rows,
    float *out, void *stream)
{
    MVF_REQUIRE(C >= 1 && nseg >= 0 && rows >= 0, "gather_csr: bad shape");
    if (nseg == 0 || rows == 0) return MVF_OK;
    MVF_REQUIRE(src && ptr && out, "gather_csr: null pointer");
    int vec, lpr_log;
    pick_layout(src, out, C, vec, lpr_log);
    const i64 threads = nseg << lpr_log;
    dim3 grid(cdiv(threads, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4)
        hipLaunchKernelGGL((gather_csr_kernel<4>), grid, block, 0, st, src, C, (const i64 *)ptr, (i64)nseg, lpr_log, out);
    else
        hipLaunchKernelGGL((gather_csr_kernel<1>), grid, block, 0, st, src, C, (const i64 *)ptr, (i64)nseg, lpr_log, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_segment_softmax_csr(
    const float *src, int32_t G, const int64_t *ptr, int64_t nseg, float eps,
    int32_t scaling, float *out, void *stream)
{
    MVF_REQUIRE(G >= 1 && nseg >= 0, "segment_softmax_csr: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(src && ptr && out, "segment_softmax_csr: null pointer");
    const i64 threads = nseg * G;
    hipLaunchKernelGGL(segment_softmax_kernel, dim3(cdiv(threads, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, G, (const i64 *)ptr, (i64)nseg, eps, scaling, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_segment_softmax_csr_backward(
    const float *grad_out, const float *y, int32_t G, const int64_t *ptr, int64_t nseg,
    int32_t scaling, float *grad_src, void *stream)
{
    MVF_REQUIRE(G >= 1 && nseg >= 0, "segment_softmax_csr_backward: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(grad_out && y && ptr && grad_src, "segment_softmax_csr_backward: null pointer");
    hipLaunchKernelGGL(segment_softmax_backward_kernel, dim3(cdiv(nseg * G, 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, y, G, (const i64 *)ptr, (i64)nseg, scaling, grad_src);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_gather_pool(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *images,
    const int64_t *view_ptr, const int16_t *pixels, int64_t rows, int32_t reduce, float *out,
    void *stream)
{
    MVF_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && rows >= 0, "gather_pool: bad shape");
    MVF_REQUIRE(reduce >= 0 && reduce <= 3, "gather_pool: unknown reduce %d", reduce);
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(x && images && view_ptr && pixels && out, "gather_pool: null pointer");
    dim3 grid(cdiv(rows, 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MVF_GP(M_) hipLaunchKernelGGL((gather_pool_kernel<M_>), grid, block, 0, st, x, C, H, W, \
                                      (const i64 *)images, (const i64 *)view_ptr, pixels, (i64)rows, out)
    switch (reduce) {
    case MVF_REDUCE_SUM: MVF_GP(MVF_REDUCE_SUM); break;
    case MVF_REDUCE_MEAN: MVF_GP(MVF_REDUCE_MEAN); break;
    case MVF_REDUCE_MIN: MVF_GP(MVF_REDUCE_MIN); break;
    default: MVF_GP(MVF_REDUCE_MAX); break;
    }
#undef MVF_GP
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_gather_features(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *batch,
    const int16_t *pixels, int64_t rows, float *out, void *stream)
{
    MVF_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && rows >= 0, "gather_features: bad shape");
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(x && batch && pixels && out, "gather_features: null pointer");
    hipLaunchKernelGGL(gather_features_kernel, dim3(cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream,
                       x, C, H, W, (const i64 *)batch, pixels, (i64)rows, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_sparse_interpolation(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const float *coords,
    const int64_t *batch, int64_t rows, int32_t padding_mode, float *out, void *stream)
{
    MVF_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && rows >= 0, "sparse_interpolation: bad shape");
    MVF_REQUIRE(padding_mode >= PAD_BORDER && padding_mode <= PAD_REFLECTION,
                "sparse_interpolation: unknown padding mode %d", padding_mode);
    MVF_REQUIRE(padding_mode != PAD_REFLECTION || (H >= 2 && W >= 2),
                "sparse_interpolation: reflection padding needs H, W >= 2");
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(x && coords && batch && out, "sparse_interpolation: null pointer");
    hipLaunchKernelGGL(sparse_interp_kernel, dim3(cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream,
                       x, C, H, W, coords, (const i64 *)batch, (i64)rows, padding_mode, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_gather_features_backward(
    const float *grad_out, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *batch,
    const int16_t *pixels, int64_t rows, float *grad_x, void *stream)
{
    MVF_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && rows >= 0, "gather_features_backward: bad shape");
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(grad_out && batch && pixels && grad_x, "gather_features_backward: null pointer");
    hipLaunchKernelGGL(scatter_features_kernel, dim3(cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, C, H, W, (const i64 *)batch, pixels, (i64)rows, grad_x);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_sparse_interpolation_backward(
    const float *grad_out, int32_t B, int32_t C, int32_t H, int32_t W, const float *coords,
    const int64_t *batch, int64_t rows, int32_t padding_mode, float *grad_x, void *stream)
{
    MVF_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && rows >= 0, "sparse_interpolation_backward: bad shape");
    MVF_REQUIRE(padding_mode >= PAD_BORDER && padding_mode <= PAD_REFLECTION,
                "sparse_interpolation_backward: unknown padding mode %d", padding_mode);
    MVF_REQUIRE(padding_mode != PAD_REFLECTION || (H >= 2 && W >= 2),
                "sparse_interpolation_backward: reflection padding needs H, W >= 2");
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(grad_out && coords && batch && grad_x, "sparse_interpolation_backward: null pointer");
    hipLaunchKernelGGL(sparse_interp_backward_kernel, dim3(cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, C, H, W, coords, (const i64 *)batch, (i64)rows, padding_mode, grad_x);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_segment_csr_two_level(
    const float *src, int64_t rows, int32_t C, const int64_t *inner_ptr, int64_t n_inner,
    int32_t inner_reduce, const int64_t *outer_ptr, int64_t n_outer, int32_t reduce_a, float *out_a,
    int32_t reduce_b, float *out_b, void *stream)
{
    MVF_REQUIRE(rows >= 0 && C >= 1 && n_inner >= 0 && n_outer >= 0, "segment_csr_two_level: bad shape");
    MVF_REQUIRE(inner_reduce >= 0 && inner_reduce <= 3 && reduce_a >= 0 && reduce_a <= 3 &&
                reduce_b >= 0 && reduce_b <= 3, "segment_csr_two_level: unknown reduce");
    if (n_outer == 0) return MVF_OK;
    MVF_REQUIRE(inner_ptr && outer_ptr && out_a && (rows == 0 || src), "segment_csr_two_level: null pointer");
    int vec = C % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out_a % 16 == 0 &&
              (!out_b || (uintptr_t)out_b % 16 == 0) ? 4 : 1;
    dim3 grid(cdiv(n_outer * (C / vec), 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MVF_TL(V_, M_) \
    hipLaunchKernelGGL((two_level_pool_kernel<V_, M_>), grid, block, 0, st, src, C, (const i64 *)inner_ptr, \
                       (const i64 *)outer_ptr, (i64)n_outer, reduce_a, out_a, reduce_b, out_b)
#define MVF_TL_MODE(V_)                                            \
    switch (inner_reduce) {                                        \
    case MVF_REDUCE_SUM: MVF_TL(V_, MVF_REDUCE_SUM); break;        \
    case MVF_REDUCE_MEAN: MVF_TL(V_, MVF_REDUCE_MEAN); break;      \
    case MVF_REDUCE_MIN: MVF_TL(V_, MVF_REDUCE_MIN); break;        \
    default: MVF_TL(V_, MVF_REDUCE_MAX); break;                    \
    }
    if (vec == 4) { MVF_TL_MODE(4) } else { MVF_TL_MODE(1) }
#undef MVF_TL_MODE
#undef MVF_TL
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
