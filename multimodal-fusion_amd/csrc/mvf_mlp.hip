// Dense pieces of the attentive pools (modules/multimodal/pooling.py:159-551,
// 604-727): chains of Linear + eval-mode BatchNorm + LeakyReLU layers on the
// fp32 matrix cores, the query/key softmax and the attention-weighted segment
// sum.
//
// MLP chain.  A wave owns 32 rows and keeps their activations in registers
// from the first load to the final store; LDS only holds the weights, shared
// by the workgroup.  Every layer is computed transposed, Y^T = W X^T, with
// v_mfma_f32_32x32x2_f32: the A operand is a weight (lane l: output channel
// l & 31 of the block, k-half l >> 5, read from LDS), the B operand is an
// activation of the lane's own row (row l & 31), and the 16 accumulators of a
// lane are 16 output channels of that same row (channel (r & 3) + 8 (r >> 2)
// + 4 (l >> 5) of the block).  An accumulator is therefore directly the B
// operand of the next layer: the order in which the input channels are summed
// is chosen to match what each lane already holds, so no transposition and no
// LDS round trip happens between layers.  For the first layer a lane reads a
// contiguous half of its row's chunk (k-half 0: first half of the columns,
// k-half 1: second half).  The instruction is an exact fp32 fmaf chain; only
// the summation order differs from a sequential dot product.
// BatchNorm (eval) and the bias collapse into one scale and one shift per
// output channel: the scale is multiplied into the weights, the shift is added to
// the finished sums (mlp_late4), the activation follows on the accumulators.
//
// Limits of one launch: every layer's output width <= 64 and every layer
// after the first has input width <= 64; the first layer's input is streamed
// in chunks of 64 columns and may be any width.  The host splits wider layers
// into several launches.
#include <stdlib.h>

#include "mvf_common.h"

namespace mvf {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifdef MVF_MLP_DEBUG
static int g_mlp_debug;          // host side; bit 0: no output stores, bit 1: no input loads
#define MLPD(A_, bit) ((A_).debug & (bit))
#define MLP_DBG(A_) ((A_).debug)
#else
#define MLPD(A_, bit) false
#define MLP_DBG(A_) 0          // a compile-time zero: the phase switches vanish from the layer loops
#endif

#ifdef MVF_CHAIN_STAMP
__device__ unsigned long long g_chain_span[2048 * 2];   // per workgroup: first / last wall clock (100 MHz)
#endif
constexpr int MLP_THREADS = 256;
constexpr int MLP_WAVES = 4;
constexpr int MLP_ROWS = 32 * MLP_WAVES;   // rows per workgroup step
constexpr int MLP_KC = 64;                 // input columns per chunk
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
    int vec_in, vec_out;  // 16-byte loads / stores are possible
    int debug;            // phase switches of MVF_MLP_DEBUG builds
    i64 rows;
    mvf_mlp_layer L[MLP_MAX_LAYERS];
    float *out;
    int ldo;
    // dot epilogue (dot_out != null): instead of the last layer's output y [rows][cout <= 32],
    // dot_out[row] = sum_c y[row][c] * dot_q[dot_seg[row]][c]  (/ dot_div when dot_div != 0)
    const float *dot_q;
    const int *dot_seg;
    int dot_ldq;
    float dot_div;
    float *dot_out;
    // pool epilogue (pool_out != null; POOL kernels): the last layer's output y is not stored;
    // pool_out[s][c] = sum over the rows r of segment s, in order, of y[r][c] * pool_att[r][group(c)]
    // with s = dot_seg[r]; rows are grouped by segment, segments ascending.  pool_att is
    // [rows][pool_groups] and already carries the gate factor of the row's segment (a
    // per-segment factor read at every segment change would be a dependent load in the
    // epilogue).  POOL == 2: pool_groups > 1 groups of cout / pool_groups channels.
    const float *pool_att;
    float *pool_out;
    int pool_groups;
    // sums (POOL 1 / 2): per-tile partial sums of the segments that cross a tile boundary,
    // [tile][2][cout]: slot 0 = the segment that began in an earlier tile and ends in this
    // one, slot 1 = the segment that continues in the next tile (whether or not it began
    // in this one).  pool_fixup_kernel adds them up in tile order.
    float *pool_part;
};

// Rows of one staging step of the pool epilogue and the LDS pitch of a staged row.
constexpr int POOL_STAGE_ROWS = 16;
constexpr int POOL_PITCH = MLP_MAX_C + 4;
// Segments are folded inside tiles of this many rows (one wave's rows); a segment that
// crosses a tile boundary leaves one partial sum per tile, which a second launch adds up
// in tile order (maxima: combined with atomic maxima, which do not depend on order).
constexpr int POOL_TILE = 32;
// Most attention groups the pool epilogue takes (their weights of a tile are parked in the
// staging rows, two groups per lane and load).
constexpr int POOL_MAX_GROUPS = 8;

__host__ __device__ __forceinline__ int pad32(int v) { return (v + 31) & ~31; }
// LDS rows of a layer's transposed weights
__host__ __device__ __forceinline__ int mlp_wrows(int layer, int cin)
{
    if (layer == 0) return cin < MLP_KC ? ((cin + 7) & ~7) : MLP_KC;   // whole groups of 4 steps per k-half
    return pad32(cin);
}

// Issue priority of a wave by its progress through its tiles.  The arbiter of a SIMD serves
// priority before age, and age alone lets the workgroup that reached a CU first finish its
// (equal) share in two thirds of the time of the one that came last (137 against 191 us in a
// launch of 203), after which the CU runs on with ever fewer waves; a wave that is a quarter
// of its share ahead now yields to the others.  Worth about 2 % per launch: with the finish
// times evened out (166 .. 194 us) the launch is hardly shorter, the CU as a whole gets
// through its tiles at the same rate either way.  (Also tried: tiles handed out on demand by
// ticket counters shared between workgroups -- 5 % on the long launches with 32 queues, but
// the returning atomics of 4096 waves on a few addresses cost the short launches 40 us, and
// with enough counters to avoid that nothing is left of the balance; larger shares for the
// workgroups that arrive first -- no change on top of the priorities.)
__device__ __forceinline__ void prio_by_progress(unsigned done, unsigned share)
{
    const unsigned quarter = share ? (4u * done) / share : 3u;
    if (quarter == 0) __builtin_amdgcn_s_setprio(3);
    else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
    else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
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

__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float4 add4(const float4 &x, const float4 &y)
{
    return make_float4(__fadd_rn(x.x, y.x), __fadd_rn(x.y, y.y), __fadd_rn(x.z, y.z), __fadd_rn(x.w, y.w));
}

// The lane's S (a multiple of 4, at most 4 XQ <= 32) input columns c0 .. c0 + S of
// its row as XQ 16-byte loads without any branch: columns beyond S are read
// from c0 again and never used.  pa / pb point at the lane's row in the two
// sources (pb = pa when there is one source); ca = columns of the first one.
template <int MODE, int XQ>
__device__ __forceinline__ void mlp_load_row(
    const float *pa, const float *pb, int ca, int c0, int S, float (&xr)[4 * XQ], bool skip)
{
    float4 u[XQ], w[XQ];
    if (skip) {
#pragma unroll
        for (int t = 0; t < 4 * XQ; ++t) xr[t] = 1.f;
        return;
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        const int c = c0 + (4 * q < S ? 4 * q : 0);
        if (MODE == 0) {
            u[q] = ld4(c < ca ? pa + c : pb + (c - ca));
        } else if (MODE == 1) {
            u[q] = ld4(pa + c);
            w[q] = ld4(pb + c);
        } else {
            const int cc = c < ca ? c : c - ca;
            u[q] = ld4(pa + cc);
            w[q] = ld4(pb + cc);
        }
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        float4 v = u[q];
        if (MODE == 1) v = add4(u[q], w[q]);
        if (MODE == 2) {
            const int c = c0 + (4 * q < S ? 4 * q : 0);
            if (c >= ca) v = add4(u[q], w[q]);
        }
        xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
    }
}

// Channel of accumulator register r inside its block of 32, for k-half kh.
__device__ __forceinline__ int acc_channel(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// Scale, shift, activation on the accumulators of output block nb.
__device__ __forceinline__ void mlp_epilogue(
    floatx16 &acc, int nb, const float *sc, const float *sh, const mvf_mlp_layer &L, int kh)
{
    const float *s = sc + nb * 32 + 4 * kh, *t = sh + nb * 32 + 4 * kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2);
        float v = __fadd_rn(__fmul_rn(acc[r], s[ch]), t[ch]);
        if (L.act) v = v > 0.f ? v : __fmul_rn(v, L.slope);
        acc[r] = v;
    }
}

// Activation of four neighbouring accumulators (registers 4 g .. 4 g + 3), in
// place.  The scale is already in (the weights in LDS carry the channel scale); the shift
// has been added by mlp_late4 or was the accumulators' start (mlp_init_acc).
__device__ __forceinline__ void mlp_epilogue4(floatx16 &acc, int g, const mvf_mlp_layer &L)
{
    // leaky(v) = max(v, slope * v) for every slope <= 1 (the host rejects larger ones);
    // no activation is slope 1.  One straight line of code for every layer: a branch on the
    // layer's settings here makes the compiler copy the whole accumulator block around it.
    const float s = L.act ? L.slope : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * g + e] = fmaxf(acc[4 * g + e], __fmul_rn(acc[4 * g + e], s));
}

// The channel shifts of a layer join its finished sums, four registers of block nb at a time
// (float32 chain and the narrow three-layer split-bf16 chain; the accumulators start from
// zero).  Started from the shift instead, a sum of few terms rounds every partial sum at the
// shift's magnitude: with 2 to 8 (live) input columns that was 1.3 to 1.8 times as far from the
// float64 value as the reference's order (sum, then scale and shift), and the pools' scores,
// gates and softmax behind it magnify the difference.  From 32 input columns on the two orders
// measure the same, which is why the wide two-layer split-bf16 chain (64 columns per layer,
// vector issue to spare for nothing) keeps the start from the shift (mlp_init_acc).
__device__ __forceinline__ void mlp_late4(floatx16 &acc, int nb, int g, const float *sh, int kh)
{
    const float4 v = *(const float4 *)(sh + nb * 32 + 4 * kh + 8 * g);
    acc[4 * g] = __fadd_rn(acc[4 * g], v.x); acc[4 * g + 1] = __fadd_rn(acc[4 * g + 1], v.y);
    acc[4 * g + 2] = __fadd_rn(acc[4 * g + 2], v.z); acc[4 * g + 3] = __fadd_rn(acc[4 * g + 3], v.w);
}

// Accumulators of block nb start from the channel shifts of the layer.
__device__ __forceinline__ void mlp_init_acc(floatx16 &acc, int nb, const float *sh, int kh)
{
    const float *t = sh + nb * 32 + 4 * kh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *(const float4 *)(t + 8 * g);
        acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
    }
}

// Store of the four finished channels of group g of block nb.
__device__ __forceinline__ void mlp_store4(
    const floatx16 &acc, int nb, int g, const MlpArgs &A, bool vec_out, int cout, i64 row, int kh)
{
    float *o = A.out + row * A.ldo + nb * 32 + 4 * kh + 8 * g;
    const int c = nb * 32 + 4 * kh + 8 * g;
    if (MLPD(A, 1)) return;
    if (vec_out) {
        if (c < cout) *(float4 *)o = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < cout) o[e] = acc[4 * g + e];
    }
}

// One input block (32 channels held in acc, block nb_in of the previous layer)
// of the next layer: the previous layer's epilogue is applied four registers at
// a time right before the MFMAs that consume them, so its vector work hides
// between matrix instructions.  PASS 0: n0 and n1 together; 1: n0 only (acc is
// finished in place); 2: n1 only from the finished acc, and after every group
// the caller's `between(g)` runs (epilogue + store of a finished n0 group).
// late: the previous layer's shifts, added to its sums here (mlp_late4).
template <int PASS, class F>
__device__ __forceinline__ void mlp_next_block(
    floatx16 &acc, int nb_in, const float *wp, int cop2, bool wide_out,
    const mvf_mlp_layer &Lprev, floatx16 &n0, floatx16 &n1, F between, int dbg,
    const float *late, int kh)
{
    const float *w = wp + nb_in * 32 * cop2;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float w0[4], w1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (PASS != 2) w0[e] = (dbg & 4) ? 0.5f : w[(8 * g + e) * cop2];
            if (PASS == 2 || (PASS == 0 && wide_out)) w1[e] = (dbg & 4) ? 0.5f : w[(8 * g + e) * cop2 + 32];
        }
        if (PASS != 2) mlp_late4(acc, nb_in, g, late, kh);
        if (PASS != 2 && !(dbg & 8)) mlp_epilogue4(acc, g, Lprev);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (PASS != 2) n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], acc[4 * g + e], n0, 0, 0, 0);
            if (PASS == 2 || (PASS == 0 && wide_out))
                n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], acc[4 * g + e], n1, 0, 0, 0);
        }
        between(nb_in * 4 + g);
    }
}

__device__ __forceinline__ void mlp_store(
    const floatx16 &acc, int nb, const MlpArgs &A, bool vec_out, int cout, i64 row, int kh)
{
    float *o = A.out + row * A.ldo + nb * 32 + 4 * kh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = nb * 32 + 4 * kh + 8 * g;
        if (vec_out) {
            if (c < cout) *(float4 *)(o + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < cout) o[8 * g + e] = acc[4 * g + e];
        }
    }
}

// max(*p, v) on a float in global memory without a compare-and-swap loop: non-negative
// floats order like signed integers, negative ones like unsigned integers reversed
// (-inf, the preset of the pool epilogue, is below every value in both views).
__device__ __forceinline__ void atomic_max_float(float *p, float v)
{
    // by the sign bit, not by value: -0.0 belongs to the negative branch
    if (!(__float_as_uint(v) >> 31)) atomicMax((int *)p, __float_as_int(v));
    else atomicMin((unsigned *)p, __float_as_uint(v));
}

// Segment reduction of a wave's 32 finished rows (acc0 / acc1: the two blocks of 32 output
// channels, lane = row jrow, channels acc_channel(r, kh)): the attention-weighted sum
// (POOL 1 / 2, pooling.py:519-530) or the maximum (POOL 3, pooling.py:628).
// The rows go through the wave's staging rows `stg` 16 at a time: a lane writes the 32 / 64
// channels of its row (for the sum: times the row's attention weight), then lane c folds
// channel c over the staged rows in row order.  The segment of a row is wave-uniform (read
// from the row's lane), so segment changes are scalar branches.  A segment that lies inside
// the tile is stored; one that continues in a neighbouring tile leaves its partial sum in
// the tile's slot of A.pool_part (the maximum: combined atomically with its row preset to
// -inf).  pseg: the row's segment (-1 past the end), pprev / pnext: the segments of the rows
// next to the tile, patt: the row's attention weight (POOL 1); POOL 2 finds the tile's
// [32][groups] weights parked in `stg`.  t0: first row of the tile.
template <int POOL>
__device__ __forceinline__ void mlp_pool_tail(
    const MlpArgs &A, floatx16 &acc0, floatx16 &acc1, bool wide, int cout, float *stg,
    int pseg, int pprev, int pnext, float patt, int lane, i64 t0)
{
    const int jrow = lane & 31, kh = lane >> 5;
    // rows times their attention weights, in place (POOL == 2: one weight per float4
    // of channels, picked up from the parked [32][groups] block)
    if (POOL == 2) {
        // c0 / (cout / groups) as a multiply and shift, exact for c0 < 64
        const unsigned inv = (65536u + (unsigned)(cout / A.pool_groups) - 1u) / (unsigned)(cout / A.pool_groups);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c0 = (q >> 2) * 32 + 8 * (q & 3) + 4 * kh;
            if (q >= 4 && !wide) break;
            const float w = c0 < cout ? stg[jrow * A.pool_groups + (int)(((unsigned)c0 * inv) >> 16)] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q < 4) acc0[4 * (q & 3) + e] = __fmul_rn(acc0[4 * (q & 3) + e], w);
                else acc1[4 * (q & 3) + e] = __fmul_rn(acc1[4 * (q & 3) + e], w);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if (POOL == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = __fmul_rn(acc0[r], patt);
            if (wide) acc1[r] = __fmul_rn(acc1[r], patt);
        }
    }
    int cur = __builtin_amdgcn_readlane(pseg, 0);
    bool partial = cur == pprev;                 // the current segment began before this tile
    // POOL == 3 folds with max instead of a sum (first row wins ties, as in the
    // segment kernels); a partial maximum meets the others in a row preset to -inf
    constexpr float FOLD0 = POOL == 3 ? -INFINITY : 0.f;
    float accv = FOLD0;
    auto flush = [&](int seg, float v, bool part, int slot) {
        if (seg < 0 || lane >= cout) return;
        float *o = A.pool_out + (size_t)seg * cout + lane;
        if (!part) *o = v;
        else if (POOL == 3) atomic_max_float(o, v);
        else A.pool_part[((size_t)(t0 / POOL_TILE) * 2 + slot) * cout + lane] = v;
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((jrow >> 4) == half) {
            float *p = stg + (jrow & 15) * POOL_PITCH + 4 * kh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4 *)(p + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
            if (wide) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4 *)(p + 32 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float v[POOL_STAGE_ROWS];
#pragma unroll
        for (int i = 0; i < POOL_STAGE_ROWS; ++i) v[i] = stg[i * POOL_PITCH + lane];
#pragma unroll
        for (int i = 0; i < POOL_STAGE_ROWS; ++i) {
            const int r = half * POOL_STAGE_ROWS + i;
            const int s = __builtin_amdgcn_readlane(pseg, r);
            if (s != cur) {
                flush(cur, accv, partial, 0);
                partial = false;
                cur = s;
                accv = FOLD0;
            }
            if (POOL == 3) accv = v[i] > accv ? v[i] : accv;
            else accv = __fadd_rn(accv, v[i]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // rows past the end carry segment -1: a tile that holds some has flushed its
    // last segment at the change and ends with cur = -1
    flush(cur, accv, partial || cur == pnext, cur == pnext ? 1 : 0);
}

// The same reduction without LDS and without a serial walk over the rows: a segmented
// inclusive scan over the lanes.  A lane's row jrow knows how far it is from the start of its
// segment inside the tile (d, from a ballot of the segment starts); in steps of 1, 2, 4 and 8
// lanes inside a DPP row of 16, and one step from lane 15 of the first row to the second row,
// every lane whose segment reaches that far back adds the partial of the lane behind it,
// for all of its 16 / 32 channel registers.  The last row of a segment then holds the
// segment's result and writes it with 16-byte stores; a segment that continues in a
// neighbouring tile leaves its partial sum in A.pool_part as in mlp_pool_tail.  The sum
// of a segment is a balanced tree over its rows instead of the in-order fold (deterministic,
// same error bound, different last bits for segments of three and more rows).  Sums only
// (POOL 1 / 2).  Needs cout % 4 == 0 and a 16-byte aligned output.  `stage2`: the wave's
// staging rows; POOL 2 finds the tile's [32][groups] attention weights parked there.
template <int POOL>
__device__ __forceinline__ void mlp_pool_tail_scan(
    const MlpArgs &A, floatx16 &acc0, floatx16 &acc1, bool wide, int cout, float *stage2,
    int pseg, int pprev, int pnext, float patt, int lane, i64 t0)
{
    static_assert(POOL == 1 || POOL == 2, "sums only");
    const float *att_lds = stage2;
    const int jrow = lane & 31, kh = lane >> 5;
    if (POOL == 2) {
        const unsigned inv = (65536u + (unsigned)(cout / A.pool_groups) - 1u) / (unsigned)(cout / A.pool_groups);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c0 = (q >> 2) * 32 + 8 * (q & 3) + 4 * kh;
            if (q >= 4 && !wide) break;
            const float w = c0 < cout ? att_lds[jrow * A.pool_groups + (int)(((unsigned)c0 * inv) >> 16)] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q < 4) acc0[4 * (q & 3) + e] = __fmul_rn(acc0[4 * (q & 3) + e], w);
                else acc1[4 * (q & 3) + e] = __fmul_rn(acc1[4 * (q & 3) + e], w);
            }
        }
        // the parked weights are overwritten by the next tile: every lane is done with them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if (POOL == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = __fmul_rn(acc0[r], patt);
            if (wide) acc1[r] = __fmul_rn(acc1[r], patt);
        }
    }
    // segment structure of the tile (the same in both lane halves)
    const int behind = __shfl_up(pseg, 1, 64);
    const bool is_start = jrow == 0 || pseg != behind;
    const unsigned starts = (unsigned)__ballot(is_start);            // lanes 0 .. 31
    const int first = 31 - __clz((int)(starts & (0xffffffffu >> (31 - jrow))));
    const int d = jrow - first;                                      // rows of the segment behind this one
    const int dd = d < (jrow & 15) ? d : (jrow & 15);                // ... inside the DPP row of 16 lanes
    const bool is_end = jrow == 31 || ((starts >> (jrow + 1)) & 1u);
    // One step for all registers: v += partial of the lane `shift` behind, times the lane's 0 / 1
    // mask, as one v_fmac_f32 with a DPP source (times 1 and plus 0 are exact).  No branch: a
    // DPP read from a lane that is switched off returns nothing, so every lane stays active.
    // A DPP read needs two wait states after a vector write of the register: inside a step the
    // instructions touch different registers, and the s_nop in front of a step covers the
    // writes of the step (or the multiplication) before it.
#define MVF_SCAN_STEP(COND_, DPP_)                                                                \
    {                                                                                             \
        const float m_ = (COND_) ? 1.f : 0.f;                                                     \
        asm volatile("s_nop 1" ::: "memory");                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                          \
            asm volatile("v_fmac_f32_dpp %0, %0, %1 " DPP_ : "+v"(acc0[r]) : "v"(m_));            \
            if (wide) asm volatile("v_fmac_f32_dpp %0, %0, %1 " DPP_ : "+v"(acc1[r]) : "v"(m_));  \
        }                                                                                         \
    }
    MVF_SCAN_STEP(dd >= 1, "row_shr:1 row_mask:0xf bank_mask:0xf")
    MVF_SCAN_STEP(dd >= 2, "row_shr:2 row_mask:0xf bank_mask:0xf")
    MVF_SCAN_STEP(dd >= 4, "row_shr:4 row_mask:0xf bank_mask:0xf")
    MVF_SCAN_STEP(dd >= 8, "row_shr:8 row_mask:0xf bank_mask:0xf")
    MVF_SCAN_STEP((jrow & 16) && d > (jrow & 15), "row_bcast:15 row_mask:0xa bank_mask:0xf")
#undef MVF_SCAN_STEP
    // Segments that continue in a neighbouring tile (at most the first and the last one, known
    // wave-wide) leave their partial sums in the tile's two slots of the partial buffer.
    const int seg0 = __builtin_amdgcn_readlane(pseg, 0), seg31 = __builtin_amdgcn_readlane(pseg, 31);
    const bool part1 = seg31 >= 0 && seg31 == pnext;
    const bool part0 = seg0 >= 0 && seg0 == pprev && !(part1 && seg0 == seg31);
    const bool mine_partial = (part0 && pseg == seg0) || (part1 && pseg == seg31);
    if (is_end && pseg >= 0) {
        float *o = mine_partial
            ? A.pool_part + ((size_t)(t0 / POOL_TILE) * 2 + (pseg == seg31 && part1 ? 1 : 0)) * cout + 4 * kh
            : A.pool_out + (size_t)pseg * cout + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (8 * g + 4 * kh < cout)
                *(float4 *)(o + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
            if (wide && 32 + 8 * g + 4 * kh < cout)
                *(float4 *)(o + 32 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
        }
    }
}

// (Built with -DMVF_SEGMAX_SCAN only: measured on 2 M rows, 8 -> 32 -> 32, the launch takes 138 us
// with this scan and 135 us with the LDS fold of mlp_pool_tail<3>, which stays the default.)
// The segment maximum (POOL 3) as a segmented inclusive scan over the lanes, like the sums of
// mlp_pool_tail_scan: in steps of 1, 2, 4, 8 lanes inside a DPP row and one step into the
// second row, every lane takes the maximum of its value and that of the lane `shift` behind,
// where its segment reaches that far back.  Per register and step: v_add_f32 with a DPP
// source and the lane's bias (0 where the step applies, -inf where it does not: the sum is
// then -inf and loses; a lane without a source reads 0, or keeps a preset -inf) and v_max_f32.  No
// branch and no exec masking: every lane stays a valid DPP source.  A maximum does not
// depend on the order of its operands, so the result has the bits of the in-order fold.
// The last row of a segment stores its channels with 16-byte stores; a segment that
// continues in a neighbouring tile meets its other parts in a row preset to -inf
// (segmax_prepare_kernel) through atomic maxima, issued from two staging rows (`stage2`) so
// that one instruction takes 64 channels.  Needs cout % 4 == 0 and a 16-byte aligned output.
__device__ __forceinline__ void mlp_pool_tail_scanmax(
    const MlpArgs &A, floatx16 &acc0, floatx16 &acc1, bool wide, int cout, float *stage2,
    int pseg, int pprev, int pnext, int lane)
{
    const int jrow = lane & 31, kh = lane >> 5;
    const int behind = __shfl_up(pseg, 1, 64);
    const bool is_start = jrow == 0 || pseg != behind;
    const unsigned starts = (unsigned)__ballot(is_start);            // lanes 0 .. 31
    const int first = 31 - __clz((int)(starts & (0xffffffffu >> (31 - jrow))));
    const int d = jrow - first;                                      // rows of the segment behind this one
    const int dd = d < (jrow & 15) ? d : (jrow & 15);                // ... inside the DPP row of 16 lanes
    const bool is_end = jrow == 31 || ((starts >> (jrow + 1)) & 1u);
    // PRESET_: the step leaves some lanes unwritten (row_mask), so the sum starts from -inf;
    // the row_shr steps read 0 for a lane without a source (bound_ctrl), whose bias is -inf.
#define MVF_SCANMAX_STEP(COND_, DPP_, PRESET_)                                                    \
    {                                                                                             \
        const float b_ = (COND_) ? 0.f : -INFINITY;                                               \
        asm volatile("s_nop 1" ::: "memory");                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                          \
            float t_, u_;                                                                         \
            if (PRESET_) {                                                                        \
                t_ = -INFINITY;                                                                   \
                asm volatile("v_add_f32_dpp %0, %1, %2 " DPP_ : "+v"(t_) : "v"(acc0[r]), "v"(b_)); \
            } else {                                                                              \
                asm volatile("v_add_f32_dpp %0, %1, %2 " DPP_ : "=v"(t_) : "v"(acc0[r]), "v"(b_)); \
            }                                                                                     \
            asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc0[r]) : "v"(t_));                       \
            if (wide) {                                                                           \
                if (PRESET_) {                                                                    \
                    u_ = -INFINITY;                                                               \
                    asm volatile("v_add_f32_dpp %0, %1, %2 " DPP_ : "+v"(u_) : "v"(acc1[r]), "v"(b_)); \
                } else {                                                                          \
                    asm volatile("v_add_f32_dpp %0, %1, %2 " DPP_ : "=v"(u_) : "v"(acc1[r]), "v"(b_)); \
                }                                                                                 \
                asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc1[r]) : "v"(u_));                   \
            }                                                                                     \
        }                                                                                         \
    }
    MVF_SCANMAX_STEP(dd >= 1, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", false)
    MVF_SCANMAX_STEP(dd >= 2, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", false)
    MVF_SCANMAX_STEP(dd >= 4, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", false)
    MVF_SCANMAX_STEP(dd >= 8, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", false)
    MVF_SCANMAX_STEP((jrow & 16) && d > (jrow & 15), "row_bcast:15 row_mask:0xa bank_mask:0xf", true)
#undef MVF_SCANMAX_STEP
    // Segments that continue in a neighbouring tile (at most the first and the last one, known
    // wave-wide) go through two staging rows so that one atomic instruction takes 64 channels.
    const int seg0 = __builtin_amdgcn_readlane(pseg, 0), seg31 = __builtin_amdgcn_readlane(pseg, 31);
    const bool part1 = seg31 >= 0 && seg31 == pnext;
    const bool part0 = seg0 >= 0 && seg0 == pprev && !(part1 && seg0 == seg31);
    const bool mine_partial = (part0 && pseg == seg0) || (part1 && pseg == seg31);
    if (is_end && pseg >= 0) {
        float *o = mine_partial ? stage2 + (pseg == seg31 && part1 ? POOL_PITCH : 0) + 4 * kh
                                : A.pool_out + (size_t)pseg * cout + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (8 * g + 4 * kh < cout)
                *(float4 *)(o + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
            if (wide && 32 + 8 * g + 4 * kh < cout)
                *(float4 *)(o + 32 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
        }
    }
    if (part0 || part1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (part0 && lane < cout) atomic_max_float(A.pool_out + (size_t)seg0 * cout + lane, stage2[lane]);
        if (part1 && lane < cout) atomic_max_float(A.pool_out + (size_t)seg31 * cout + lane, stage2[POOL_PITCH + lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// NL / WMASK >= 0 fix the number of layers and which of them are wider than 32
// outputs (bit l of WMASK) at compile time, for the aligned single-source or
// concatenated inputs (mode 0, 16-byte loads and stores): layer loops unroll
// and LDS offsets become immediates.  NL < 0 is the general kernel that takes
// everything from the arguments.
// XQ: 16-byte pieces of the row a lane holds for the first layer (8 = any width
// up to 64 columns; narrow first layers take fewer registers and loads).
template <int NL, int WMASK, int POOL, int XQ = 8>
__device__ __forceinline__ void mlp_chain_body(const MlpArgs &A)
{
    constexpr bool FIXED = NL >= 0;
    static_assert(FIXED || XQ == 8, "the general kernel holds whole chunks");
    const int nlayers = FIXED ? NL : A.nlayers;
    const int mode = FIXED ? 0 : A.mode;
    const bool vec_in = FIXED ? true : A.vec_in != 0;
    const bool vec_out = FIXED ? true : A.vec_out != 0;
    // padded output width of layer l
    auto cop_of = [&](int l) { return FIXED ? (((WMASK >> l) & 1) ? 64 : 32) : pad32(A.L[l].cout); };
#ifdef MVF_CHAIN_STAMP
    const long long cstart_ = clock64();
    const unsigned long long wstart_ = wall_clock64();
#endif
    extern __shared__ float smem[];
    float *sc = smem;                                   // [3][64] channel scales
    float *sh = sc + MLP_MAX_LAYERS * MLP_MAX_C;        // [3][64] channel shifts
    // transposed weights [k][cout_pad] of the layers, as offsets into smem (plain
    // offsets keep the accesses LDS instructions)
    int woff[MLP_MAX_LAYERS];
    int stage_off;                                      // POOL: staging rows of the waves, behind the weights
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jrow = lane & 31, kh = lane >> 5;

    const int cin0 = A.L[0].cin;
    const int nchunks = (cin0 + MLP_KC - 1) / MLP_KC;
    {
        int w = 2 * MLP_MAX_LAYERS * MLP_MAX_C;
#pragma unroll
        for (int l = 0; l < MLP_MAX_LAYERS; ++l) {
            woff[l] = w;
            if (l < nlayers) w += mlp_wrows(l, A.L[l].cin) * cop_of(l);
        }
        stage_off = w;
    }
    // channel scale / shift: y = (x W^T + bias - mean) / sqrt(var + eps) * gamma + beta
    for (int idx = tid; idx < nlayers * MLP_MAX_C; idx += MLP_THREADS) {
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
    __syncthreads();
    // resident weights (times the channel scale): every layer but a chunked first
    // one; rows and columns beyond the layer are zero
    for (int l = (nchunks > 1 ? 1 : 0); l < nlayers; ++l) {
        const mvf_mlp_layer &L = A.L[l];
        const int kin = mlp_wrows(l, L.cin), cop = cop_of(l);
        const int cshift = cop > 32 ? 6 : 5, total = (kin >> 2) * cop;     // kin is a multiple of 8
        const int wbase = l == 0 ? woff[0] : l == 1 ? woff[1] : woff[2];
        // An item is four consecutive input columns of one output channel: one 16-byte load
        // where the layer allows it, and the four LDS stores of a wave go to consecutive
        // addresses.  (Every workgroup of the launch reads the same few kilobytes at the same
        // time: element-wise reads, 64 cache lines per wave instruction, queue up at one L2
        // channel.)  Four items of a thread are in flight together.
        const bool vec = (L.cin & 3) == 0 && ((uintptr_t)L.weight & 15) == 0;
        for (int i0 = tid; i0 < total; i0 += 4 * MLP_THREADS) {
            float4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * MLP_THREADS, k = (idx >> cshift) << 2, j = idx & (cop - 1);
                w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total && j < L.cout) {
                    const float *src = L.weight + (size_t)j * L.cin + k;
                    if (vec && k + 3 < L.cin) w[u] = ld4(src);
                    else {
                        if (k < L.cin) w[u].x = src[0];
                        if (k + 1 < L.cin) w[u].y = src[1];
                        if (k + 2 < L.cin) w[u].z = src[2];
                        if (k + 3 < L.cin) w[u].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * MLP_THREADS, k = (idx >> cshift) << 2, j = idx & (cop - 1);
                if (idx < total) {
                    const float sj = sc[l * MLP_MAX_C + j];
                    float *dst = smem + wbase + k * cop + j;
                    dst[0] = __fmul_rn(w[u].x, sj);
                    dst[cop] = __fmul_rn(w[u].y, sj);
                    dst[2 * cop] = __fmul_rn(w[u].z, sj);
                    dst[3 * cop] = __fmul_rn(w[u].w, sj);
                }
            }
        }
    }
    __syncthreads();

    const int cop0 = cop_of(0);
    float xr[4 * XQ];
    // The lane's half of a 64-column chunk of its row, 16-byte loads.  Rows past
    // the end read row 0; their results are never stored.
    // Source rows of the lane's row of the tile at base_ (row gathers applied).  Rows past
    // the end read row 0; their results are never stored.
    // (the pool kernels take ungathered rows only: the host checks)
    auto gather_rows = [&](i64 base_, int &ga, int &gb) {
        const i64 row_ = base_ + wave * 32 + jrow;
        const bool live_ = row_ < A.rows;
        ga = gb = 0;
        if (!POOL && live_ && A.a.gather) ga = A.a.gather[row_];
        if (!POOL && live_ && A.b.gather) gb = A.b.gather[row_];
    };
    auto source_rows = [&](i64 base_, int ga, int gb, i64 &ra, i64 &rb) {
        const i64 row_ = base_ + wave * 32 + jrow;
        ra = rb = row_ < A.rows ? row_ : 0;
        if (!POOL && A.a.gather) ra = ga;
        if (!POOL && A.b.gather) rb = gb;
    };
    // The lane's half of a 64-column chunk of its row, 16-byte loads.
    auto load_rows_at = [&](i64 ra, i64 rb, int k0, int S) {
        const float *pa = A.a.p + ra * A.a.ld;
        const float *pb = A.b.p ? A.b.p + rb * A.b.ld : pa;
        if (mode == 0) mlp_load_row<0, XQ>(pa, pb, A.a.cols, k0 + kh * S, S, xr, MLPD(A, 2));
        else if (mode == 1) mlp_load_row<1, XQ>(pa, pb, A.a.cols, k0 + kh * S, S, xr, MLPD(A, 2));
        else mlp_load_row<2, XQ>(pa, pb, A.a.cols, k0 + kh * S, S, xr, MLPD(A, 2));
    };
    auto load_rows = [&](i64 base_, int k0, int S) {
        int ga, gb;
        i64 ra, rb;
        gather_rows(base_, ga, gb);
        source_rows(base_, ga, gb, ra, rb);
        load_rows_at(ra, rb, k0, S);
    };
    // With a single chunk the next tile's rows are requested right after the
    // first layer has consumed the current ones, so their latency hides behind
    // the remaining layers.
    const bool prefetch = vec_in && nchunks == 1;
    const i64 stride = (i64)gridDim.x * MLP_ROWS;
    if (prefetch && (i64)blockIdx.x * MLP_ROWS < A.rows) load_rows((i64)blockIdx.x * MLP_ROWS, 0, (cin0 + 1) >> 1);
    // With row gathers the rows of a tile hang on its gather indices: those are fetched one
    // tile further ahead (nga / ngb: gather indices of the tile after the current one), so
    // the request for the rows never waits for them.
    const bool gathered = !POOL && prefetch && (A.a.gather || A.b.gather);
    int nga = 0, ngb = 0;
    if (gathered && (i64)blockIdx.x * MLP_ROWS + stride < A.rows) gather_rows((i64)blockIdx.x * MLP_ROWS + stride, nga, ngb);
#ifdef MVF_CHAIN_STAMP
    long long cs_[4] = {0, 0, 0, 0}; int cn_ = 0;
    const long long cw0_ = wall_clock64(), cc0_ = clock64();
#define CSTAMP(i_) { const long long n_ = clock64(); cs_[i_] += n_ - clast_; clast_ = n_; }
#else
#define CSTAMP(i_) {}
#endif
    const unsigned my_share = (unsigned)((A.rows - (i64)blockIdx.x * MLP_ROWS + stride - 1) / stride);
    unsigned my_done = 0;
    for (i64 base = (i64)blockIdx.x * MLP_ROWS; base < A.rows; base += stride) {
        prio_by_progress(my_done++, my_share);
#ifdef MVF_CHAIN_STAMP
        long long clast_ = clock64(); ++cn_;
#endif
        const i64 row = base + wave * 32 + jrow;
        const bool live = row < A.rows;
        floatx16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;     // the shifts come at the end (mlp_late4)
        // dot epilogue: the row's query index now, its query values after the first
        // layer, so that neither load is waited for at the end of the tile
        const int qseg = (A.dot_out && live) ? A.dot_seg[row] : 0;
        float qv[16];
        // pool epilogue: the row's segment and attention weight now; the segments of the rows
        // next to the wave's tile tell whether its first / last segment continues in a
        // neighbouring tile
        int pseg = -1, pprev = -1, pnext = -1;
        float patt = 0.f;
        // POOL == 2: the [32][groups] attention weights of the wave's rows are contiguous in
        // memory; they are fetched here with coalesced loads, parked in the (idle) staging rows
        // after the first layer and picked up per lane right before the epilogue
        float pa[POOL_MAX_GROUPS / 2];
        if (POOL) {
            const i64 t0 = base + wave * 32;
            if (live) pseg = A.dot_seg[row];
            if (POOL == 1 && live) patt = A.pool_att[row];
            if (POOL == 2) {
                const i64 first = t0 * A.pool_groups, stop = A.rows * A.pool_groups;
#pragma unroll
                for (int k = 0; k < POOL_MAX_GROUPS / 2; ++k) {
                    const int idx = lane + 64 * k;
                    pa[k] = (idx < 32 * A.pool_groups && first + idx < stop) ? A.pool_att[first + idx] : 0.f;
                }
            }
            if (t0 > 0 && t0 < A.rows) pprev = A.dot_seg[t0 - 1];
            if (t0 + POOL_TILE < A.rows) pnext = A.dot_seg[t0 + POOL_TILE];
        }

        // ---- first layer: the lane's half of each 64-column chunk of its row ----
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int k0 = chunk * MLP_KC;
            const int kc = cin0 - k0 < MLP_KC ? cin0 - k0 : MLP_KC;
            const int S = (kc + 1) >> 1;                 // MFMA steps of the chunk
            if (nchunks > 1) {
                __syncthreads();                         // every wave is done with the previous chunk
                const mvf_mlp_layer &L = A.L[0];
                for (int idx = tid; idx < MLP_KC * cop0; idx += MLP_THREADS) {
                    const int k = idx / cop0, j = idx - k * cop0;
                    smem[woff[0] + idx] = (k < kc && j < L.cout) ? __fmul_rn(L.weight[(size_t)j * L.cin + k0 + k], sc[j]) : 0.f;
                }
            }
            const int c_lo = kh * S;                     // first column of this lane inside the chunk
            if (!prefetch) {
                if (vec_in) {
                    load_rows(base, k0, S);
                } else {
                    i64 ra = row, rb = row;
                    if (live && A.a.gather) ra = A.a.gather[row];
                    if (live && A.b.gather) rb = A.b.gather[row];
#pragma unroll
                    for (int t = 0; t < 4 * XQ; ++t) {
                        xr[t] = 0.f;
                        if (live && t < S && c_lo + t < kc) xr[t] = mlp_input(A, ra, rb, k0 + c_lo + t);
                    }
                }
            }
            if (nchunks > 1) __syncthreads();
            // groups of 4 steps: the weights of a group are read together; steps
            // beyond S meet zero activations (scalar path) or are whole groups
            const float *wp = smem + woff[0] + c_lo * cop0 + jrow;
            if (cop0 > 32) {
#pragma unroll
                for (int g = 0; g < XQ; ++g)
                    if (4 * g < S) {
                        float w0[4], w1[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { w0[e] = MLPD(A, 4) ? 0.5f : wp[(4 * g + e) * cop0]; w1[e] = MLPD(A, 4) ? 0.5f : wp[(4 * g + e) * cop0 + 32]; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], xr[4 * g + e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], xr[4 * g + e], acc1, 0, 0, 0);
                        }
                    }
            } else {
#pragma unroll
                for (int g = 0; g < XQ; ++g)
                    if (4 * g < S) {
                        float w0[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w0[e] = wp[(4 * g + e) * cop0];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], xr[4 * g + e], acc0, 0, 0, 0);
                    }
            }
        }
        CSTAMP(0)
        if (POOL) {
            // The pool's per-row loads of this tile are waited for here, before the next
            // tile's rows are requested: that request is conditional, so any later wait for
            // a load older than it would have to wait for it as well.
            if (POOL == 2) {
                float *stg_att = smem + stage_off + wave * (POOL_STAGE_ROWS * POOL_PITCH);
#pragma unroll
                for (int k = 0; k < POOL_MAX_GROUPS / 2; ++k)
                    if (lane + 64 * k < 32 * A.pool_groups) stg_att[lane + 64 * k] = pa[k];
            }
            asm volatile("" : "+v"(pseg), "+v"(patt), "+v"(pprev), "+v"(pnext) : : "memory");
        }
        if (A.dot_out) {
            const float *qrow = A.dot_q + (size_t)qseg * A.dot_ldq;
            const int cq = A.L[nlayers - 1].cout;
            // 16-byte loads (the host checks alignment and cout % 4 == 0)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * kh + 8 * g;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < cq) t = ld4(qrow + c);
                qv[4 * g] = t.x; qv[4 * g + 1] = t.y; qv[4 * g + 2] = t.z; qv[4 * g + 3] = t.w;
            }
        }
        // The next tile's rows, requested behind every load of this tile that depends on an
        // earlier one (the request is conditional: a later wait for an older load would wait
        // for it too).  (prefetch implies a single chunk, whose steps are S = (cin0 + 1) / 2)
        if (prefetch && base + stride < A.rows) {
            if (gathered) {
                i64 ra, rb;
                source_rows(base + stride, nga, ngb, ra, rb);
                load_rows_at(ra, rb, 0, (cin0 + 1) >> 1);
                if (base + 2 * stride < A.rows) gather_rows(base + 2 * stride, nga, ngb);
            } else {
                load_rows(base + stride, 0, (cin0 + 1) >> 1);
            }
        }
        // ---- the following layers, activations stay in registers ----
        auto nothing = [](int) {};
#pragma unroll
        for (int l = 0; l + 1 < (FIXED ? NL : MLP_MAX_LAYERS); ++l) {
            if (l + 1 >= nlayers) break;
            const mvf_mlp_layer &L = A.L[l];
            const bool wide_in = cop_of(l) > 32, wide_out = cop_of(l + 1) > 32;
            const int cop2 = cop_of(l + 1);
            // the lane's register r of block nb is input channel nb * 32 + acc_channel(r, kh)
            const float *wp = smem + (l == 0 ? woff[1] : woff[2]) + (4 * kh) * cop2 + jrow;
            const float *late = sh + l * MLP_MAX_C;     // layer l takes its shifts here
            floatx16 n0, n1;
#pragma unroll
            for (int r = 0; r < 16; ++r) n0[r] = n1[r] = 0.f;
            if (l + 2 == nlayers && wide_out) {
                // last layer with two output blocks: n0 first, then n1 while the
                // finished n0 is scaled and stored group by group
                const mvf_mlp_layer &L2 = A.L[l + 1];
                mlp_next_block<1>(acc0, 0, wp, cop2, true, L, n0, n1, nothing, MLP_DBG(A), late, kh);
                if (wide_in) mlp_next_block<1>(acc1, 1, wp, cop2, true, L, n0, n1, nothing, MLP_DBG(A), late, kh);
                const int per = wide_in ? 2 : 1;      // n1 groups per n0 group
                auto drain = [&](int step) {
                    if ((step + 1) % per == 0) {
                        const int g = step / per;
                        mlp_late4(n0, 0, g, sh + (l + 1) * MLP_MAX_C, kh);
                        mlp_epilogue4(n0, g, L2);
                        if ((POOL == 0 || POOL == 3) && live) mlp_store4(n0, 0, g, A, vec_out, L2.cout, row, kh);
                    }
                };
                mlp_next_block<2>(acc0, 0, wp, cop2, true, L, n0, n1, drain, MLP_DBG(A), late, kh);
                if (wide_in) mlp_next_block<2>(acc1, 1, wp, cop2, true, L, n0, n1, drain, MLP_DBG(A), late, kh);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mlp_late4(n1, 1, g, sh + (l + 1) * MLP_MAX_C, kh);
                    mlp_epilogue4(n1, g, L2);
                    if ((POOL == 0 || POOL == 3) && live) mlp_store4(n1, 1, g, A, vec_out, L2.cout, row, kh);
                }
                acc0 = n0; acc1 = n1;
                if (POOL) goto pool_tail;
                goto tile_done;
            }
            mlp_next_block<0>(acc0, 0, wp, cop2, wide_out, L, n0, n1, nothing, MLP_DBG(A), late, kh);
            if (wide_in) mlp_next_block<0>(acc1, 1, wp, cop2, wide_out, L, n0, n1, nothing, MLP_DBG(A), late, kh);
            acc0 = n0;
            acc1 = n1;
        }
        CSTAMP(1)
        if (A.dot_out) {
            // The last layer's rows meet one query row each and only the products'
            // sum leaves the kernel (the keys and compatibilities of the attentive
            // pools, pooling.py:503-508).  A row's 32 channels sit in the two lanes
            // jrow and jrow + 32.
            const mvf_mlp_layer &L = A.L[nlayers - 1];
            float part = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mlp_late4(acc0, 0, g, sh + (nlayers - 1) * MLP_MAX_C, kh);
                mlp_epilogue4(acc0, g, L);
                const int c = 4 * kh + 8 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < L.cout) part = __fadd_rn(part, __fmul_rn(acc0[4 * g + e], qv[4 * g + e]));
            }
            part = __fadd_rn(part, __shfl_xor(part, 32, 64));
            if (live && kh == 0) A.dot_out[row] = A.dot_div != 0.f ? __fdiv_rn(part, A.dot_div) : part;
        } else {
            // epilogue and store of the last layer
            const mvf_mlp_layer &L = A.L[nlayers - 1];
            const bool wide = cop_of(nlayers - 1) > 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mlp_late4(acc0, 0, g, sh + (nlayers - 1) * MLP_MAX_C, kh);
                mlp_epilogue4(acc0, g, L);
                if ((POOL == 0 || POOL == 3) && live) mlp_store4(acc0, 0, g, A, vec_out, L.cout, row, kh);
            }
            if (wide) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mlp_late4(acc1, 1, g, sh + (nlayers - 1) * MLP_MAX_C, kh);
                    mlp_epilogue4(acc1, g, L);
                    if ((POOL == 0 || POOL == 3) && live) mlp_store4(acc1, 1, g, A, vec_out, L.cout, row, kh);
                }
            }
        }
    pool_tail:
#ifdef MVF_SEGMAX_SCAN
        if (POOL == 3)
            mlp_pool_tail_scanmax(A, acc0, acc1, cop_of(nlayers - 1) > 32, A.L[nlayers - 1].cout,
                                  smem + stage_off + wave * (POOL_STAGE_ROWS * POOL_PITCH),
                                  pseg, pprev, pnext, lane);
        else
#endif
        if (POOL)
            mlp_pool_tail<POOL>(A, acc0, acc1, cop_of(nlayers - 1) > 32, A.L[nlayers - 1].cout,
                                smem + stage_off + wave * (POOL_STAGE_ROWS * POOL_PITCH),
                                pseg, pprev, pnext, patt, lane, base + wave * 32);
    tile_done:;
        CSTAMP(2)
    }
#ifdef MVF_CHAIN_STAMP
    if (tid == 0 && blockIdx.x < 2048) { g_chain_span[blockIdx.x * 2] = wstart_; g_chain_span[blockIdx.x * 2 + 1] = wall_clock64(); }
    if (blockIdx.x == 9 && tid == 0 && cn_ > 0)
        printf("chain NL %d grid %d tiles %d: first layer %lld | next layers %lld | epilogue %lld cycles per tile, whole loop %lld, prologue %lld, %.0f MHz\n", NL, (int)gridDim.x, cn_,
               cs_[0] / cn_, cs_[1] / cn_, cs_[2] / cn_, clock64() - cc0_, cc0_ - cstart_, (double)(clock64() - cc0_) / (double)(wall_clock64() - cw0_) * 100.0);
#endif
}

template <int NL, int WMASK, int XQ = 8>
__global__ __launch_bounds__(MLP_THREADS) void mlp_chain_kernel(MlpArgs A)
{
    mlp_chain_body<NL, WMASK, 0, XQ>(A);
}

// The pool epilogue's staging rows leave LDS for three workgroups per CU; the register
// budget is held to the same three waves per SIMD (the scheduler otherwise spends registers
// on early weight reads and drops to two).
template <int NL, int WMASK, int POOL, int XQ = 8>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(3)))
void mlp_chain_pool_kernel(MlpArgs A)
{
    mlp_chain_body<NL, WMASK, POOL, XQ>(A);
}

// ------------------------------------------------------------------------- //
// wide two-layer chains on the bf16 matrix cores, float32 accuracy           //
// ------------------------------------------------------------------------- //
// v_mfma_f32_32x32x16_bf16 runs at 16 times the rate of the float32 form.  A float32 value
// splits exactly into three bf16 parts, x = x1 + x2 + x3 (each the round-to-nearest bf16 of
// what the parts before it leave; the remainders are exact float32 subtractions), and the
// product of two split values is the sum of nine part products that are exact in float32.
// The six of weight 2^-16 and above -- (1,1) (1,2) (2,1) (2,2) (1,3) (3,1), added smallest
// first -- leave out less than 2^-23 of the product, the rounding of a float32 multiply: six
// bf16 MFMAs do the work of eight float32 MFMAs in 3/8 of the matrix-pipe time, with the
// accuracy of the float32 chain (measured against float64 on random layers: 3.4e-6 against
// 4.6e-6 for a float32 dot product at magnitude 13).  Nothing overflows or loses range: bf16
// has float32's exponent.
//
// The price is vector work: 11 instructions per pair of values to split (v_cvt_pk_bf16_f32,
// two shifts / masks, two subtractions, twice over, and a last conversion), 60 per k-step of 8
// values with the activation.  A wave issues in order and an MFMA holds the vector issue port
// for 8 of its 32 cycles, so the split of the NEXT k-step is written between the MFMAs of the
// current one, five instructions per gap, in stage-major order (neighbouring instructions do
// not depend on each other) and fenced with scheduling barriers so that the order survives
// the compiler.  One wave per SIMD (one workgroup per CU): two such streams on one SIMD only
// take matrix-pipe slots from each other.
//
// Layout as in the float32 chain: Y^T = W X^T, a lane owns row jrow = lane & 31, the 16
// accumulators of a block are the next layer's B operand without any lane movement.  The k
// order of a step follows what a lane holds: first layer, k-step q, lane half kh, element j
// = input column kh * S + 8 q + j (S = 8 * k-steps: the lane's contiguous half of its row);
// later layers, k-step q = registers 8 (q & 1) .. + 7 of block q >> 1, i.e. input channel
// 32 (q >> 1) + acc_channel(8 (q & 1) + j, kh).  The weights sit in LDS already multiplied
// by the channel scale, split, and in that order: [layer][k-step][out block][part][lane]
// x 16 bytes, one conflict-free ds_read_b128 per operand, requested one k-step ahead.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define MVF_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)
{
    floatx2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// v_max_f32 as written: fmaxf() first passes an operand of unknown origin (an MFMA result)
// through another v_max_f32 to quiet a signalling NaN, one more instruction per value.
__device__ __forceinline__ float max_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The split of 8 values (after the activation of the layer that produced them) as a list of
// single vector instructions in stage-major order; 44 operations, 60 with the activation.
struct Bf3Split {
    float x[8], t[8];
    unsigned d[4], h[4], g[4];
    uintx4 p1, p2, p3;
    __device__ __forceinline__ void op(int i, const float *v, bool act, float slope)
    {
        if (act) {
            // leaky(v) = max(v, slope v) for slope <= 1; no activation is slope 1
            if (i < 8) { t[i] = __fmul_rn(v[i], slope); return; }
            if (i < 16) { x[i - 8] = max_raw(v[i - 8], t[i - 8]); return; }
            i -= 16;
        }
        const int st = i >> 2, e = i & 3;
        switch (st) {
        case 0:
            if (!act) { x[2 * e] = v[2 * e]; x[2 * e + 1] = v[2 * e + 1]; }
            d[e] = cvt_pk_bf16(x[2 * e], x[2 * e + 1]); p1[e] = d[e];
            break;
        case 1: case 6: h[e] = d[e] << 16; break;
        case 2: case 7: g[e] = d[e] & 0xffff0000u; break;
        case 3: case 8: x[2 * e] = __fsub_rn(x[2 * e], __uint_as_float(h[e])); break;
        case 4: case 9: x[2 * e + 1] = __fsub_rn(x[2 * e + 1], __uint_as_float(g[e])); break;
        case 5: d[e] = cvt_pk_bf16(x[2 * e], x[2 * e + 1]); p2[e] = d[e]; break;
        default: p3[e] = cvt_pk_bf16(x[2 * e], x[2 * e + 1]); break;
        }
    }
    // piece i of n equal pieces of the list
    __device__ __forceinline__ void piece(int i, int n, const float *v, bool act, float slope)
    {
        const int total = act ? 60 : 44, per = (total + n - 1) / n;
#pragma unroll
        for (int k = i * per; k < (i + 1) * per; ++k)
            if (k < total) op(k, v, act, slope);
    }
    __device__ __forceinline__ void all(const float *v, bool act, float slope) { piece(0, 1, v, act, slope); }
};

__device__ __forceinline__ floatx16 mfma_bf16(const uintx4 &a, const uintx4 &b, floatx16 acc)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// One k-step into both output blocks: 12 MFMAs, after each of them piece i of `fill`.  The
// weights of this step are in `a` (block 0 parts 1..3, block 1 parts 1..3); those of the next
// step, at `nxt`, are requested first and replace them at the end.
template <class F>
__device__ __forceinline__ void bf3_kstep(
    uintx4 (&a)[6], const uintx4 *nxt, const uintx4 (&b)[3], floatx16 &n0, floatx16 &n1, F fill)
{
    uintx4 t[6];
    MVF_SB();
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] = nxt[i * 64];
    MVF_SB();
    n0 = mfma_bf16(a[0], b[2], n0); MVF_SB(); fill(0); MVF_SB();
    n0 = mfma_bf16(a[2], b[0], n0); MVF_SB(); fill(1); MVF_SB();
    n0 = mfma_bf16(a[1], b[1], n0); MVF_SB(); fill(2); MVF_SB();
    n0 = mfma_bf16(a[0], b[1], n0); MVF_SB(); fill(3); MVF_SB();
    n0 = mfma_bf16(a[1], b[0], n0); MVF_SB(); fill(4); MVF_SB();
    n0 = mfma_bf16(a[0], b[0], n0); MVF_SB(); fill(5); MVF_SB();
    n1 = mfma_bf16(a[3], b[2], n1); MVF_SB(); fill(6); MVF_SB();
    n1 = mfma_bf16(a[5], b[0], n1); MVF_SB(); fill(7); MVF_SB();
    n1 = mfma_bf16(a[4], b[1], n1); MVF_SB(); fill(8); MVF_SB();
    n1 = mfma_bf16(a[3], b[1], n1); MVF_SB(); fill(9); MVF_SB();
    n1 = mfma_bf16(a[4], b[0], n1); MVF_SB(); fill(10); MVF_SB();
    n1 = mfma_bf16(a[3], b[0], n1); MVF_SB(); fill(11); MVF_SB();
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = t[i];
}

// 16-byte vectors of one layer's LDS image with nq k-steps
__host__ __device__ __forceinline__ int bf3_layer_vecs(int nq) { return nq * 2 * 3 * 64; }

// Two layers, both wider than 32 outputs; first layer of at most 64 input columns (a multiple
// of 4, one source or two concatenated ones, 16-byte aligned rows, no row gathers).  NQ0:
// k-steps of the first layer (16 input columns each).  POOL as in mlp_chain_body.
template <int NQ0, int POOL>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_wide2_bf3_kernel(MlpArgs A)
{
    extern __shared__ float smem[];
    float *sc = smem;                                   // [2][64] channel scales
    float *sh = sc + 2 * MLP_MAX_C;                     // [2][64] channel shifts
    uintx4 *wl = (uintx4 *)(smem + 4 * MLP_MAX_C);      // the split weights of both layers
    float *stage = (float *)(wl + bf3_layer_vecs(NQ0) + bf3_layer_vecs(4));
    constexpr int S = 8 * NQ0;                          // columns of a lane half in the first layer
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jrow = lane & 31, kh = lane >> 5;
    const int cin0 = A.L[0].cin;
    for (int idx = tid; idx < 2 * MLP_MAX_C; idx += MLP_THREADS) {
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
    __syncthreads();
    // split weights (times the channel scale), zero beyond the layer's rows and columns
    for (int idx = tid; idx < (NQ0 + 4) * 2 * 64; idx += MLP_THREADS) {
        const int ln = idx & 63, nbo = (idx >> 6) & 1, qq = idx >> 7;
        const int l = qq >= NQ0, q = l ? qq - NQ0 : qq;
        const mvf_mlp_layer &L = A.L[l];
        const int m = nbo * 32 + (ln & 31), h = ln >> 5;
        // the 8 input channels of the item are two runs of 4 consecutive ones: 16-byte loads
        // where the layer allows it (every workgroup reads the same weights at the same time)
        float v[8];
        const bool vec = (L.cin & 3) == 0 && ((uintptr_t)L.weight & 15) == 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = 8 * (q & 1) + 4 * half;
            const int k = l ? 32 * (q >> 1) + acc_channel(r, h) : h * S + 8 * q + 4 * half;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < L.cout) {
                const float *src = L.weight + (size_t)m * L.cin + k;
                if (vec && k + 3 < L.cin) t = ld4(src);
                else {
                    if (k < L.cin) t.x = src[0];
                    if (k + 1 < L.cin) t.y = src[1];
                    if (k + 2 < L.cin) t.z = src[2];
                    if (k + 3 < L.cin) t.w = src[3];
                }
            }
            const float sm = sc[l * MLP_MAX_C + m];
            v[4 * half] = __fmul_rn(t.x, sm); v[4 * half + 1] = __fmul_rn(t.y, sm);
            v[4 * half + 2] = __fmul_rn(t.z, sm); v[4 * half + 3] = __fmul_rn(t.w, sm);
        }
        Bf3Split sp;
        sp.all(v, false, 1.f);
        uintx4 *dst = wl + (l ? bf3_layer_vecs(NQ0) : 0) + ((q * 2 + nbo) * 3) * 64 + ln;
        dst[0] = sp.p1; dst[64] = sp.p2; dst[128] = sp.p3;
    }
    __syncthreads();

    const i64 stride = (i64)gridDim.x * MLP_ROWS;
    i64 base = (i64)blockIdx.x * MLP_ROWS;
    if (base >= A.rows) return;
    // Piece q (8 columns) of the lane's half of its row in tile `b`; columns past the row's
    // end re-read column 0 (their weights are zero), rows past the end re-read row 0.
    float xr[4 * 2 * NQ0], xn[4 * 2 * NQ0];              // this tile's rows, the next tile's
    auto load_piece = [&](float *dst, i64 b, int q) {
        i64 r = b + wave * 32 + jrow;
        if (r >= A.rows) r = 0;
        const float *pa = A.a.p + r * A.a.ld;
        const float *pb = A.b.p ? A.b.p + r * A.b.ld : pa;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int c = kh * S + 8 * q + 4 * u;
            if (c >= cin0) c = 0;
            const float4 t = ld4(c < A.a.cols ? pa + c : pb + (c - A.a.cols));
            dst[8 * q + 4 * u] = t.x; dst[8 * q + 4 * u + 1] = t.y; dst[8 * q + 4 * u + 2] = t.z; dst[8 * q + 4 * u + 3] = t.w;
        }
    };
#pragma unroll
    for (int q = 0; q < NQ0; ++q) load_piece(xr, base, q);
    Bf3Split sp;
    uintx4 cur[3];                                      // the split k-step about to be multiplied
    sp.all(xr, false, 1.f);
    cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
    const mvf_mlp_layer &L0 = A.L[0], &L1 = A.L[1];
    const float slope0 = L0.act ? L0.slope : 1.f;
    uintx4 aw[6];                                       // weights of the k-step about to run
#pragma unroll
    for (int i = 0; i < 6; ++i) aw[i] = wl[i * 64 + lane];
    // no load is in flight when the tile loop is entered, as at its back edge (see the wait in
    // front of the last k-step): a wait inside the loop for the first tile's rows would be a
    // wait for every load of every tile
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
#ifdef MVF_BF3_STAMP
    long long st_[16] = {0}; int nt_ = 0;
    const long long w0_ = wall_clock64(), c0_ = clock64();
#define STAMP(i_) { const long long n_ = clock64(); st_[i_] += n_ - last_; last_ = n_; }
#else
#define STAMP(i_) {}
#endif
    const unsigned my_share = (unsigned)((A.rows - base + stride - 1) / stride);
    unsigned my_done = 0;
    for (; base < A.rows; base += stride) {
        prio_by_progress(my_done++, my_share);
#ifdef MVF_BF3_STAMP
        long long last_ = clock64(); ++nt_;
#endif
        const i64 row = base + wave * 32 + jrow;
        const bool live = row < A.rows;
        // the weights are re-read for every tile (in registers they would take 192 of them)
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        const uintx4 *w0 = wl + lofs;                   // the second layer's image follows the first
        // pool epilogue: the row's segment and attention weight now (see mlp_chain_body)
        int pseg = -1, pprev = -1, pnext = -1;
        float patt = 0.f;
        float pa[POOL_MAX_GROUPS / 2];
        if (POOL) {
            const i64 t0 = base + wave * 32;
            if (live) pseg = A.dot_seg[row];
            if (POOL == 1 && live) patt = A.pool_att[row];
            if (POOL == 2) {
                const i64 first = t0 * A.pool_groups, stop = A.rows * A.pool_groups;
#pragma unroll
                for (int k = 0; k < POOL_MAX_GROUPS / 2; ++k) {
                    const int idx = lane + 64 * k;
                    pa[k] = (idx < 32 * A.pool_groups && first + idx < stop) ? A.pool_att[first + idx] : 0.f;
                }
            }
            if (t0 > 0 && t0 < A.rows) pprev = A.dot_seg[t0 - 1];
            if (t0 + POOL_TILE < A.rows) pnext = A.dot_seg[t0 + POOL_TILE];
        }
        // the next tile's rows, a whole tile ahead of their first use (the split of piece 0 at
        // the end of this tile): under load a row takes longer to arrive than one layer runs
#pragma unroll
        for (int q = 0; q < NQ0; ++q) load_piece(xn, base + stride, q);
        floatx16 acc0, acc1, n0, n1;
        mlp_init_acc(acc0, 0, sh, kh);
        mlp_init_acc(acc1, 1, sh, kh);
        mlp_init_acc(n0, 0, sh + MLP_MAX_C, kh);
        mlp_init_acc(n1, 1, sh + MLP_MAX_C, kh);
        STAMP(3)
        // ---- first layer: k-step q runs beside the split of piece q + 1, whose registers
        // are then refilled for the next tile ----
#pragma unroll
        for (int q = 0; q < NQ0; ++q) {
            if (q > 0) { STAMP(3 + q) }
            if (q + 1 < NQ0) {
                bf3_kstep(aw, w0 + ((q + 1) * 6) * 64, cur, acc0, acc1,
                          [&](int i) { sp.piece(i, 12, xr + 8 * (q + 1), false, 1.f); });
                cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
            } else {
                // last k-step: the first 8 channels of the result are not finished yet
                bf3_kstep(aw, w0 + ((q + 1) * 6) * 64, cur, acc0, acc1, [&](int) {});
            }
        }
        if (POOL == 2) {
            float *stg_att = stage + wave * (POOL_STAGE_ROWS * POOL_PITCH);
#pragma unroll
            for (int k = 0; k < POOL_MAX_GROUPS / 2; ++k)
                if (lane + 64 * k < 32 * A.pool_groups) stg_att[lane + 64 * k] = pa[k];
        }
        STAMP(0)
        // ---- second layer ----
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc0[j];
        sp.all(v8, true, slope0);
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        STAMP(8)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q > 0) { STAMP(8 + q) }
            if (q < 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = ((q + 1) >> 1) ? acc1[8 * ((q + 1) & 1) + j] : acc0[8 * ((q + 1) & 1) + j];
                bf3_kstep(aw, w0 + ((NQ0 + q + 1) * 6) * 64, cur, n0, n1,
                          [&](int i) { sp.piece(i, 12, v8, true, slope0); });
            } else {
                // Every load of this tile (the next tile's rows and the pool's per-row values,
                // requested at its start) has landed by now: waiting for all of them here costs
                // little and leaves only the epilogue's stores in flight at the next tile's
                // start.  Without it the first wait of the next tile, behind the epilogue's
                // branches, is a wait for everything including that tile's own loads.
                __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
                STAMP(12)
                // the next tile's piece 0; the weights of the first layer's first step
                bf3_kstep(aw, w0, cur, n0, n1, [&](int i) { sp.piece(i, 12, xn, false, 1.f); });
            }
            cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        }
        // the split parts are due here: without this the compiler moves the last fill, whose
        // results only the next tile uses, behind the epilogue's branches
        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]));
        STAMP(1)
        // ---- activation of the last layer, then rows out or into the segment reduction ----
        const float slope1 = L1.act ? L1.slope : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            n0[r] = max_raw(n0[r], __fmul_rn(n0[r], slope1));
            n1[r] = max_raw(n1[r], __fmul_rn(n1[r], slope1));
        }
        if ((POOL == 0 || POOL == 3) && live) {
#pragma unroll
            for (int g = 0; g < 4; ++g) mlp_store4(n0, 0, g, A, true, L1.cout, row, kh);
#pragma unroll
            for (int g = 0; g < 4; ++g) mlp_store4(n1, 1, g, A, true, L1.cout, row, kh);
        }
        if (POOL == 1 || POOL == 2)
            mlp_pool_tail_scan<POOL == 2 ? 2 : 1>(A, n0, n1, true, L1.cout, stage + wave * (POOL_STAGE_ROWS * POOL_PITCH),
                                                  pseg, pprev, pnext, patt, lane, base + wave * 32);
        else if (POOL == 3)
#ifdef MVF_SEGMAX_SCAN
            mlp_pool_tail_scanmax(A, n0, n1, true, L1.cout, stage + wave * (POOL_STAGE_ROWS * POOL_PITCH),
                                  pseg, pprev, pnext, lane);
#else
            mlp_pool_tail<3>(A, n0, n1, true, L1.cout, stage + wave * (POOL_STAGE_ROWS * POOL_PITCH),
                             pseg, pprev, pnext, patt, lane, base + wave * 32);
#endif
#pragma unroll
        for (int t = 8; t < 8 * NQ0; ++t) xr[t] = xn[t];
        STAMP(2)
    }
#ifdef MVF_BF3_STAMP
    if (blockIdx.x == 9 && tid == 0)
        printf("tiles %d: start %lld | L1 k0 %lld k1 %lld k2 %lld k3+park %lld | split %lld | L2 k0 %lld k1 %lld k2 %lld wait %lld k3 %lld | tail %lld cycles per tile, %.0f MHz\n", nt_,
               st_[3] / nt_, st_[4] / nt_, st_[5] / nt_, st_[6] / nt_, st_[0] / nt_, st_[8] / nt_, st_[9] / nt_, st_[10] / nt_, st_[11] / nt_, st_[12] / nt_, st_[1] / nt_, st_[2] / nt_,
               (double)(clock64() - c0_) / (double)(wall_clock64() - w0_) * 100.0);
#endif
}

// One k-step into one output block: 6 MFMAs, behind each of them two pieces of `fill` (the
// split of a k-step is 44 / 60 vector instructions: with 6 MFMAs to hide behind, the chain is
// bound by the vector unit, at about 1.5 times the matrix-pipe time).  `a`: parts 1..3 of this
// step's weights; those of the next step, at `nxt`, are requested first.
template <class F>
__device__ __forceinline__ void bf3_kstep1(
    uintx4 (&a)[3], const uintx4 *nxt, const uintx4 (&b)[3], floatx16 &n0, F fill)
{
    uintx4 t[3];
    MVF_SB();
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = nxt[i * 64];
    MVF_SB();
    n0 = mfma_bf16(a[0], b[2], n0); MVF_SB(); fill(0); MVF_SB();
    n0 = mfma_bf16(a[2], b[0], n0); MVF_SB(); fill(1); MVF_SB();
    n0 = mfma_bf16(a[1], b[1], n0); MVF_SB(); fill(2); MVF_SB();
    n0 = mfma_bf16(a[0], b[1], n0); MVF_SB(); fill(3); MVF_SB();
    n0 = mfma_bf16(a[1], b[0], n0); MVF_SB(); fill(4); MVF_SB();
    n0 = mfma_bf16(a[0], b[0], n0); MVF_SB(); fill(5); MVF_SB();
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = t[i];
}

// Three layers of at most 32 outputs each (the key / query chains: two encoder layers and a
// Linear) on the bf16 matrix cores with split operands, rows stored or met with their query
// row (dot epilogue, as in mlp_chain_body); first layer of at most 64 input columns from one
// or two concatenated sources, each with an optional row gather.  LDS image of the weights:
// [k-step NQ0 + 2 + 2][part 3][lane 64] x 16 bytes.
template <int NQ0>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 3)))
void mlp_narrow3_bf3_kernel(MlpArgs A)
{
    extern __shared__ float smem[];
    float *sc = smem;                                   // [3][64] channel scales
    float *sh = sc + 3 * MLP_MAX_C;                     // [3][64] channel shifts
    uintx4 *wl = (uintx4 *)(smem + 6 * MLP_MAX_C);
    constexpr int S = 8 * NQ0;
    constexpr int KSTEPS = NQ0 + 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jrow = lane & 31, kh = lane >> 5;
    const int cin0 = A.L[0].cin;
    for (int idx = tid; idx < 3 * MLP_MAX_C; idx += MLP_THREADS) {
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
    __syncthreads();
    for (int idx = tid; idx < KSTEPS * 64; idx += MLP_THREADS) {
        const int ln = idx & 63, qq = idx >> 6;
        const int l = qq < NQ0 ? 0 : qq < NQ0 + 2 ? 1 : 2;
        const int q = l == 0 ? qq : l == 1 ? qq - NQ0 : qq - NQ0 - 2;
        const mvf_mlp_layer &L = A.L[l];
        const int m = ln & 31, h = ln >> 5;
        float v[8];
        const bool vec = (L.cin & 3) == 0 && ((uintptr_t)L.weight & 15) == 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = l ? acc_channel(8 * q + 4 * half, h) : h * S + 8 * q + 4 * half;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < L.cout) {
                const float *src = L.weight + (size_t)m * L.cin + k;
                if (vec && k + 3 < L.cin) t = ld4(src);
                else {
                    if (k < L.cin) t.x = src[0];
                    if (k + 1 < L.cin) t.y = src[1];
                    if (k + 2 < L.cin) t.z = src[2];
                    if (k + 3 < L.cin) t.w = src[3];
                }
            }
            const float sm = sc[l * MLP_MAX_C + m];
            v[4 * half] = __fmul_rn(t.x, sm); v[4 * half + 1] = __fmul_rn(t.y, sm);
            v[4 * half + 2] = __fmul_rn(t.z, sm); v[4 * half + 3] = __fmul_rn(t.w, sm);
        }
        Bf3Split sp;
        sp.all(v, false, 1.f);
        uintx4 *dst = wl + (qq * 3) * 64 + ln;
        dst[0] = sp.p1; dst[64] = sp.p2; dst[128] = sp.p3;
    }
    __syncthreads();

    const i64 stride = (i64)gridDim.x * MLP_ROWS;
    i64 base = (i64)blockIdx.x * MLP_ROWS;
    if (base >= A.rows) return;
    // Source rows of the lane's row in tile `b` (rows past the end read row 0); the gather
    // indices of a tile are fetched a tile before its rows.
    auto gather_of = [&](i64 b, int &ga, int &gb) {
        const i64 r = b + wave * 32 + jrow;
        ga = gb = 0;
        if (r < A.rows && A.a.gather) ga = A.a.gather[r];
        if (r < A.rows && A.b.gather) gb = A.b.gather[r];
    };
    float xr[4 * 2 * NQ0], xn[4 * 2 * NQ0];
    auto load_piece = [&](float *dst, i64 b, int ga, int gb, int q) {
        i64 r = b + wave * 32 + jrow;
        if (r >= A.rows) r = 0;
        const float *pa = A.a.p + (A.a.gather ? (i64)ga : r) * A.a.ld;
        const float *pb = A.b.p ? A.b.p + (A.b.gather ? (i64)gb : r) * A.b.ld : pa;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int c = kh * S + 8 * q + 4 * u;
            if (c >= cin0) c = 0;
            const float4 t = ld4(c < A.a.cols ? pa + c : pb + (c - A.a.cols));
            dst[8 * q + 4 * u] = t.x; dst[8 * q + 4 * u + 1] = t.y; dst[8 * q + 4 * u + 2] = t.z; dst[8 * q + 4 * u + 3] = t.w;
        }
    };
    int ga0, gb0, ga1, gb1;                             // gather indices: this tile's (prologue only), the next tile's
    gather_of(base, ga0, gb0);
    gather_of(base + stride, ga1, gb1);
#pragma unroll
    for (int q = 0; q < NQ0; ++q) load_piece(xr, base, ga0, gb0, q);
    Bf3Split sp;
    uintx4 cur[3];
    sp.all(xr, false, 1.f);
    cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
    const mvf_mlp_layer &L0 = A.L[0], &L1 = A.L[1], &L2 = A.L[2];
    const float slope0 = L0.act ? L0.slope : 1.f, slope1 = L1.act ? L1.slope : 1.f, slope2 = L2.act ? L2.slope : 1.f;
    uintx4 aw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) aw[i] = wl[i * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): no load in flight at the loop entry
    const unsigned my_share = (unsigned)((A.rows - base + stride - 1) / stride);
    unsigned my_done = 0;
    for (; base < A.rows; base += stride) {
        prio_by_progress(my_done++, my_share);
        const i64 row = base + wave * 32 + jrow;
        const bool live = row < A.rows;
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        const uintx4 *w0 = wl + lofs;
        // dot epilogue: the row's query index and, behind it, its query values
        const int qseg = (A.dot_out && live) ? A.dot_seg[row] : 0;
        // the next tile's rows a whole tile ahead, then the gather indices of the tile after
#pragma unroll
        for (int q = 0; q < NQ0; ++q) load_piece(xn, base + stride, ga1, gb1, q);
        gather_of(base + 2 * stride, ga1, gb1);
        floatx16 acc, nacc, macc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;      // the shifts come at the end (mlp_late4)
#pragma unroll
        for (int r = 0; r < 16; ++r) nacc[r] = macc[r] = 0.f;
        // ---- first layer ----
#pragma unroll
        for (int q = 0; q < NQ0; ++q) {
            if (q + 1 < NQ0) {
                bf3_kstep1(aw, w0 + ((q + 1) * 3) * 64, cur, acc,
                           [&](int i) { sp.piece(2 * i, 12, xr + 8 * (q + 1), false, 1.f); sp.piece(2 * i + 1, 12, xr + 8 * (q + 1), false, 1.f); });
                cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
            } else {
                bf3_kstep1(aw, w0 + ((q + 1) * 3) * 64, cur, acc, [&](int) {});
            }
        }
        float qv[16];
        if (A.dot_out) {
            const float *qrow = A.dot_q + (size_t)qseg * A.dot_ldq;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * kh + 8 * g;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < L2.cout) t = ld4(qrow + c);
                qv[4 * g] = t.x; qv[4 * g + 1] = t.y; qv[4 * g + 2] = t.z; qv[4 * g + 3] = t.w;
            }
        }
        // ---- second layer: k-step 0 = registers 0 .. 7 of the first layer's block ----
#pragma unroll
        for (int g = 0; g < 4; ++g) mlp_late4(acc, 0, g, sh, kh);
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc[j];
        sp.all(v8, true, slope0);
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc[8 + j];
        bf3_kstep1(aw, w0 + ((NQ0 + 1) * 3) * 64, cur, nacc,
                   [&](int i) { sp.piece(2 * i, 12, v8, true, slope0); sp.piece(2 * i + 1, 12, v8, true, slope0); });
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        bf3_kstep1(aw, w0 + ((NQ0 + 2) * 3) * 64, cur, nacc, [&](int) {});
        // ---- third layer ----
#pragma unroll
        for (int g = 0; g < 4; ++g) mlp_late4(nacc, 0, g, sh + MLP_MAX_C, kh);
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = nacc[j];
        sp.all(v8, true, slope1);
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = nacc[8 + j];
        bf3_kstep1(aw, w0 + ((NQ0 + 3) * 3) * 64, cur, macc,
                   [&](int i) { sp.piece(2 * i, 12, v8, true, slope1); sp.piece(2 * i + 1, 12, v8, true, slope1); });
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        // every load of this tile has landed by now (see mlp_wide2_bf3_kernel)
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        bf3_kstep1(aw, w0, cur, macc,
                   [&](int i) { sp.piece(2 * i, 12, xn, false, 1.f); sp.piece(2 * i + 1, 12, xn, false, 1.f); });
        cur[0] = sp.p1; cur[1] = sp.p2; cur[2] = sp.p3;
        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]));
        // ---- activation of the last layer, then the dot product or the rows ----
#pragma unroll
        for (int g = 0; g < 4; ++g) mlp_late4(macc, 0, g, sh + 2 * MLP_MAX_C, kh);
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[r] = max_raw(macc[r], __fmul_rn(macc[r], slope2));
        if (A.dot_out) {
            float part = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * kh + 8 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < L2.cout) part = __fadd_rn(part, __fmul_rn(macc[4 * g + e], qv[4 * g + e]));
            }
            part = __fadd_rn(part, __shfl_xor(part, 32, 64));
            if (live && kh == 0) A.dot_out[row] = A.dot_div != 0.f ? __fdiv_rn(part, A.dot_div) : part;
        } else if (live) {
#pragma unroll
            for (int g = 0; g < 4; ++g) mlp_store4(macc, 0, g, A, A.vec_out != 0, L2.cout, row, kh);
        }
#pragma unroll
        for (int t = 8; t < 8 * NQ0; ++t) xr[t] = xn[t];
    }
}

static size_t bf3_narrow3_lds_bytes(int nq0)
{
    return sizeof(float) * 6 * MLP_MAX_C + 16 * (size_t)((nq0 + 4) * 3 * 64);
}

static size_t bf3_lds_bytes(int nq0, bool pool)
{
    return sizeof(float) * 4 * MLP_MAX_C + 16 * (size_t)(bf3_layer_vecs(nq0) + bf3_layer_vecs(4)) +
           (pool ? sizeof(float) * MLP_WAVES * POOL_STAGE_ROWS * POOL_PITCH : 0);
}

// ------------------------------------------------------------------------- //
// attention scores inside CSR segments                                       //
// ------------------------------------------------------------------------- //
// comp[r][g] = sum_k K[r][g D + k] * Q[s(r)][g D + k], products and sums rounded
// one by one, k ascending (the reference's elementwise product followed by
// sum(dim=2), pooling.py:504-508), optionally / sqrt(D).  One thread per
// (row, group): neighbouring threads read neighbouring keys.
__global__ __launch_bounds__(256) void attention_comp_kernel(
    const float *__restrict__ keys, int ldk, const float *__restrict__ queries, int ldq,
    const int *__restrict__ row2seg, i64 rows, int G, int D, float dim_div,
    float *__restrict__ comp)
{
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * G) return;
    const i64 r = t / G;
    const int g = (int)(t - r * G);
    const i64 qr = row2seg ? (i64)row2seg[r] : r;
    const float *k = keys + (size_t)r * ldk + (size_t)g * D;
    const float *q = queries + (size_t)qr * ldq + (size_t)g * D;
    float c;
    if ((D & 3) == 0 && (ldk & 3) == 0 && (ldq & 3) == 0 &&
        ((uintptr_t)keys & 15) == 0 && ((uintptr_t)queries & 15) == 0) {
        float4 kv = *(const float4 *)k, qv = *(const float4 *)q;
        c = __fmul_rn(kv.x, qv.x);
        c = __fadd_rn(c, __fmul_rn(kv.y, qv.y));
        c = __fadd_rn(c, __fmul_rn(kv.z, qv.z));
        c = __fadd_rn(c, __fmul_rn(kv.w, qv.w));
        for (int d = 4; d < D; d += 4) {
            kv = *(const float4 *)(k + d); qv = *(const float4 *)(q + d);
            c = __fadd_rn(c, __fmul_rn(kv.x, qv.x));
            c = __fadd_rn(c, __fmul_rn(kv.y, qv.y));
            c = __fadd_rn(c, __fmul_rn(kv.z, qv.z));
            c = __fadd_rn(c, __fmul_rn(kv.w, qv.w));
        }
    } else {
        c = __fmul_rn(k[0], q[0]);
        for (int d = 1; d < D; ++d) c = __fadd_rn(c, __fmul_rn(k[d], q[d]));
    }
    if (dim_div != 0.f) c = __fdiv_rn(c, dim_div);
    comp[t] = c;
}

// One thread per (segment, group): att = segment softmax of comp (max-centred,
// optional / sqrt(rows), eps in the denominator; pooling.py:770-822), gate[s][g]
// = tanh(relu(max_r comp * Gw[g] + Gb[g])) (pooling.py:717-723, 523-526).
__global__ __launch_bounds__(256) void attention_scores_kernel(
    const i64 *__restrict__ ptr, i64 nseg, int G, int group_scaling,
    float eps, const float *__restrict__ gate_w, const float *__restrict__ gate_b,
    const float *__restrict__ comp, float *__restrict__ att, float *__restrict__ gate,
    float *__restrict__ pool_out, int pool_C)
{
    // pool_out != null: the launch prepares the pool epilogue of the layer chain.  The
    // rows of empty segments and of segments that cross a tile boundary start from zero
    // (thread (s, g) clears the channels of group g), and the gate is folded into the
    // attention weights.
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= nseg * G) return;
    const i64 s = t / G;
    const int g = (int)(t - s * G);
    const i64 a = ptr[s], b = ptr[s + 1];
    if (pool_out && (b <= a || a / POOL_TILE != (b - 1) / POOL_TILE)) {
        // (a multiple of 4 channels per group and a 16-byte aligned output: the host checks)
        const int cg = pool_C / G;
        float4 *o = (float4 *)(pool_out + (size_t)s * pool_C + (size_t)g * cg);
        for (int c = 0; c < cg / 4; ++c) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float fold = 1.f;
    float mx = 0.f;                                    // segment max of nothing is 0
    // Segments of at most SHORT rows (most of them: a point is seen in a handful of views)
    // keep their compatibilities in registers: one round of loads, all in flight together.
    constexpr int SHORT = 8;
    const i64 n = b - a;
    float cs[SHORT];
    if (n <= SHORT) {
#pragma unroll
        for (int i = 0; i < SHORT; ++i) cs[i] = i < n ? comp[(size_t)(a + i) * G + g] : 0.f;
#pragma unroll
        for (int i = 0; i < SHORT; ++i)
            if (i < n) mx = (i == 0 || cs[i] > mx) ? cs[i] : mx;
    } else {
        // longer segments: SHORT rows per step, their loads in flight together (a loop of
        // single dependent loads makes the few lanes with many views hold up the whole launch)
        for (i64 r0 = a; r0 < b; r0 += SHORT) {
#pragma unroll
            for (int i = 0; i < SHORT; ++i) cs[i] = r0 + i < b ? comp[(size_t)(r0 + i) * G + g] : 0.f;
#pragma unroll
            for (int i = 0; i < SHORT; ++i)
                if (r0 + i < b) mx = ((r0 == a && i == 0) || cs[i] > mx) ? cs[i] : mx;
        }
    }
    if (gate) {
        float x = mx;
        if (gate_w) x = __fmul_rn(x, gate_w[g]);
        if (gate_b) x = __fadd_rn(x, gate_b[g]);
        fold = tanhf(x > 0.f ? x : 0.f);
        gate[t] = fold;
    }
    if (!(pool_out && gate)) fold = 1.f;               // times 1 is exact
    if (n <= 0) return;
    const float sq = sqrtf((float)n);
    float sum = 0.f;
    if (n <= SHORT) {
#pragma unroll
        for (int i = 0; i < SHORT; ++i)
            if (i < n) {
                float c = __fsub_rn(cs[i], mx);
                if (group_scaling) c = __fdiv_rn(c, sq);
                cs[i] = expf(c);
                sum = i == 0 ? cs[i] : __fadd_rn(sum, cs[i]);
            }
        sum = __fadd_rn(sum, eps);
#pragma unroll
        for (int i = 0; i < SHORT; ++i)
            if (i < n) att[(size_t)(a + i) * G + g] = __fmul_rn(__fdiv_rn(cs[i], sum), fold);
        return;
    }
    for (i64 r0 = a; r0 < b; r0 += SHORT) {
#pragma unroll
        for (int i = 0; i < SHORT; ++i) cs[i] = r0 + i < b ? comp[(size_t)(r0 + i) * G + g] : 0.f;
#pragma unroll
        for (int i = 0; i < SHORT; ++i)
            if (r0 + i < b) {
                float c = __fsub_rn(cs[i], mx);
                if (group_scaling) c = __fdiv_rn(c, sq);
                const float e = expf(c);
                att[(size_t)(r0 + i) * G + g] = e;
                sum = (r0 == a && i == 0) ? e : __fadd_rn(sum, e);
            }
    }
    sum = __fadd_rn(sum, eps);
    for (i64 r0 = a; r0 < b; r0 += SHORT) {
#pragma unroll
        for (int i = 0; i < SHORT; ++i) cs[i] = r0 + i < b ? att[(size_t)(r0 + i) * G + g] : 0.f;
#pragma unroll
        for (int i = 0; i < SHORT; ++i)
            if (r0 + i < b) att[(size_t)(r0 + i) * G + g] = __fmul_rn(__fdiv_rn(cs[i], sum), fold);
    }
}

// x_pool[s][c] = (sum_r x[r][c] * att[r][group(c)]) * gate[s][group(c)], rows
// in order.  The output is walked flat in vectors of VEC channels (VEC = 4
// needs every vector inside one group), four (segment, vector) pairs in flight
// per thread and four rows per step, as in segment_csr.
constexpr int ATT_ILP = 4;

template <int VEC> struct AttVec;
template <> struct AttVec<1> {
    typedef float T;
    static __device__ __forceinline__ float mul(float x, float w) { return __fmul_rn(x, w); }
    static __device__ __forceinline__ float fma_(float acc, float x, float w) { return __fadd_rn(acc, __fmul_rn(x, w)); }
    static __device__ __forceinline__ float zero() { return 0.f; }
};
template <> struct AttVec<4> {
    typedef float4 T;
    static __device__ __forceinline__ float4 mul(const float4 &x, float w)
    {
        return make_float4(__fmul_rn(x.x, w), __fmul_rn(x.y, w), __fmul_rn(x.z, w), __fmul_rn(x.w, w));
    }
    static __device__ __forceinline__ float4 fma_(const float4 &a, const float4 &x, float w)
    {
        return make_float4(__fadd_rn(a.x, __fmul_rn(x.x, w)), __fadd_rn(a.y, __fmul_rn(x.y, w)),
                           __fadd_rn(a.z, __fmul_rn(x.z, w)), __fadd_rn(a.w, __fmul_rn(x.w, w)));
    }
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};

template <int VEC>
__global__ __launch_bounds__(256) void attention_pool_kernel(
    const float *__restrict__ x, int C, const float *__restrict__ att, int G,
    const int *__restrict__ chan2group, const float *__restrict__ gate,
    const i64 *__restrict__ ptr, i64 nseg, float *__restrict__ out)
{
    typedef AttVec<VEC> AV;
    typedef typename AV::T V;
    const unsigned cv = (unsigned)(C / VEC);
    const i64 total = nseg * cv;
    const V *xs = (const V *)x;
    V *o = (V *)out;
    const i64 e0 = (i64)blockIdx.x * (256 * ATT_ILP) + threadIdx.x;
    const i64 seg0 = ((i64)blockIdx.x * (256 * ATT_ILP)) / cv;
    const unsigned rem0 = (unsigned)((i64)blockIdx.x * (256 * ATT_ILP) - seg0 * cv) + threadIdx.x;
    i64 a[ATT_ILP], b[ATT_ILP], s[ATT_ILP];
    unsigned c[ATT_ILP];
    int g[ATT_ILP];
    V acc[ATT_ILP];
#pragma unroll
    for (int k = 0; k < ATT_ILP; ++k) {
        a[k] = b[k] = s[k] = 0; c[k] = 0; g[k] = 0;
        if (e0 + k * 256 < total) {
            const unsigned off = rem0 + k * 256, ds = off / cv;
            s[k] = seg0 + ds;
            c[k] = off - ds * cv;
            g[k] = chan2group ? chan2group[c[k] * VEC] : 0;
            a[k] = ptr[s[k]];
            b[k] = ptr[s[k] + 1];
        }
    }
#pragma unroll
    for (int k = 0; k < ATT_ILP; ++k) {
        acc[k] = AV::zero();
        if (b[k] > a[k]) acc[k] = AV::mul(xs[(size_t)a[k] * cv + c[k]], att[(size_t)a[k] * G + g[k]]);
    }
#pragma unroll
    for (int k = 0; k < ATT_ILP; ++k) {
        // four rows at a time, loaded together (rows past the end predicated off),
        // accumulated in row order
        for (i64 r = a[k] + 1; r < b[k]; r += 4) {
            const V *q = xs + (size_t)r * cv + c[k];
            const float *w = att + (size_t)r * G + g[k];
            const bool v1 = r + 1 < b[k], v2 = r + 2 < b[k], v3 = r + 3 < b[k];
            const V x0 = q[0];
            const V x1 = v1 ? q[(size_t)cv] : AV::zero();
            const V x2 = v2 ? q[2 * (size_t)cv] : AV::zero();
            const V x3 = v3 ? q[3 * (size_t)cv] : AV::zero();
            const float w0 = w[0], w1 = v1 ? w[G] : 0.f, w2 = v2 ? w[2 * G] : 0.f, w3 = v3 ? w[3 * G] : 0.f;
            acc[k] = AV::fma_(acc[k], x0, w0);
            if (v1) acc[k] = AV::fma_(acc[k], x1, w1);
            if (v2) acc[k] = AV::fma_(acc[k], x2, w2);
            if (v3) acc[k] = AV::fma_(acc[k], x3, w3);
        }
        if (e0 + k * 256 < total) {
            if (gate) acc[k] = AV::mul(acc[k], gate[(size_t)s[k] * G + g[k]]);
            o[e0 + k * 256] = acc[k];
        }
    }
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
    const i64 *__restrict__ ptr, i64 nseg, int pick_max, float *__restrict__ out,
    i64 *__restrict__ picked)
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
        if (picked && c == 0) picked[s] = best;
    } else if (picked && c == 0) {
        picked[s] = -1;
    }
    out[e] = v;
}

// Ahead of the max-pool epilogue: rows of empty segments are 0 (the pools' value of an empty
// segment), rows of segments that cross a 32-row tile start from -inf; every other row is
// stored by the epilogue itself.
__global__ __launch_bounds__(256) void segmax_prepare_kernel(
    const i64 *__restrict__ ptr, i64 nseg, float *__restrict__ out, int C)
{
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const i64 a = ptr[s], b = ptr[s + 1];
    if (b > a && a / POOL_TILE == (b - 1) / POOL_TILE) return;
    const float v = b > a ? -INFINITY : 0.f;
    float4 *o = (float4 *)(out + (size_t)s * C);
    for (int c = 0; c < C / 4; ++c) o[c] = make_float4(v, v, v, v);
}

static size_t mlp_lds_bytes(const mvf_mlp_layer *L, int n)
{
    size_t f = 2 * MLP_MAX_LAYERS * MLP_MAX_C;
    for (int l = 0; l < n; ++l) f += (size_t)mlp_wrows(l, L[l].cin) * pad32(L[l].cout);
    return f * sizeof(float);
}

}  // namespace mvf

using namespace mvf;

// 1: wide layer chains run on the bf16 matrix cores with three-way split operands
// (float32 accuracy); 0: every chain on the float32 matrix cores (exact fmaf chains).
static int g_mlp_precision = [] {
    const char *e = getenv("MVF_MLP_PRECISION");
    return (e && (e[0] == 'f' || e[0] == '0')) ? 0 : 1;      // "fp32" / "0" select the float32 cores
}();

extern "C" int mvf_set_mlp_precision(int32_t mode)
{
    MVF_REQUIRE(mode == 0 || mode == 1, "set_mlp_precision: mode %d (0 = float32 matrix cores, 1 = split bf16)", mode);
    g_mlp_precision = mode;
    return MVF_OK;
}

extern "C" int mvf_get_mlp_precision(void) { return g_mlp_precision; }

#ifdef MVF_CHAIN_STAMP
extern "C" int mvf_debug_chain_span(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_span), sizeof(unsigned long long) * 2 * n) != hipSuccess;
}
#endif

#ifdef MVF_MLP_DEBUG
extern "C" int mvf_debug_mlp_switch(int mask)
{
    g_mlp_debug = mask;
    return 0;
}
#endif

static int mlp_forward_impl(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    float *out, int32_t ldo, const float *dot_q, int32_t dot_ldq, const int32_t *dot_seg,
    float dot_div, float *dot_out, void *stream,
    const float *pool_att = nullptr, float *pool_out = nullptr, int32_t pool_groups = 1,
    bool pool_max = false, float *pool_part = nullptr)
{
    MVF_REQUIRE(rows >= 0 && layers && nlayers >= 1 && nlayers <= MLP_MAX_LAYERS,
                "mlp_forward: bad arguments (nlayers=%d)", nlayers);
    MVF_REQUIRE(mode >= 0 && mode <= 2, "mlp_forward: unknown input mode %d", mode);
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(xa && cols_a >= 1 && lda >= cols_a && (out || dot_out || pool_out), "mlp_forward: bad first input");
    if (pool_out && pool_max) {
        MVF_REQUIRE(dot_seg && !dot_out && out && !gather_a && !gather_b,
                    "mlp_forward_segmax: needs the row output and the row -> segment map, no row gathers");
    } else if (pool_out) {
        MVF_REQUIRE(pool_att && dot_seg && !dot_out && !out && pool_part,
                    "mlp_forward_pool: needs the attention weights, the row -> segment map and the partial-sum buffer");
        MVF_REQUIRE(!gather_a && !gather_b, "mlp_forward_pool: row gathers are not supported");
        const int co = layers[nlayers - 1].cout;
        static_assert(32 * POOL_MAX_GROUPS <= POOL_STAGE_ROWS * POOL_PITCH, "the weights of a tile are parked in its staging rows");
        MVF_REQUIRE(pool_groups >= 1 && pool_groups <= POOL_MAX_GROUPS &&
                    (pool_groups == 1 || (co % pool_groups == 0 && (co / pool_groups) % 4 == 0)),
                    "mlp_forward_pool: %d groups; at most 8 equal groups of a multiple of 4 channels", pool_groups);
        out = pool_out; ldo = layers[nlayers - 1].cout;                 // never written through `out`
    }
    if (dot_out) {
        MVF_REQUIRE(dot_q && dot_seg && layers[nlayers - 1].cout <= 32 && dot_ldq >= layers[nlayers - 1].cout,
                    "mlp_forward_dot: needs query rows, the row -> query map and at most 32 output channels");
        MVF_REQUIRE(layers[nlayers - 1].cout % 4 == 0 && dot_ldq % 4 == 0 && (uintptr_t)dot_q % 16 == 0,
                    "mlp_forward_dot: query rows must be 16-byte aligned, with a multiple of 4 channels");
        if (!out) { out = dot_out; ldo = layers[nlayers - 1].cout; }    // never written; keeps the checks below simple
    }
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
        MVF_REQUIRE(!layers[l].act || layers[l].slope <= 1.f,
                    "mlp_forward: layer %d has a LeakyReLU slope %g > 1", l, (double)layers[l].slope);
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
    A.dot_q = dot_q; A.dot_seg = dot_seg; A.dot_ldq = dot_ldq; A.dot_div = dot_div; A.dot_out = dot_out;
    A.pool_att = pool_att; A.pool_out = pool_out; A.pool_groups = pool_groups; A.pool_part = pool_part;
    if (pool_out) A.dot_seg = dot_seg;
    A.debug = 0;
#ifdef MVF_MLP_DEBUG
    A.debug = g_mlp_debug;
#endif
    // 16-byte access: every chunk splits into halves of whole float4s that stay inside one source
    const bool al_a = ((uintptr_t)xa % 16 == 0) && lda % 4 == 0;
    const bool al_b = !xb || (((uintptr_t)xb % 16 == 0) && ldb % 4 == 0);
    A.vec_in = al_a && al_b && cin0 % 8 == 0 && cols_a % 4 == 0;
    A.vec_out = dot_out || (pool_out && !pool_max) || (((uintptr_t)out % 16 == 0) && ldo % 4 == 0 && layers[nlayers - 1].cout % 4 == 0);
    const size_t lds = mlp_lds_bytes(layers, nlayers) +
                       (pool_out ? sizeof(float) * MLP_WAVES * POOL_STAGE_ROWS * POOL_PITCH : 0);
    // compute units of the current device (looked up once per device)
    static int cu_cache[64];
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_cache[dev] == 0) {
            hipDeviceProp_t prop;
            cu_cache[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0
                                ? prop.multiProcessorCount : 256;
        }
        cus = cu_cache[dev];
    }
    i64 per_cu = (i64)(160 * 1024 / (lds + 1024));
    if (per_cu > 8) per_cu = 8;
    int forced_per_cu = 0;
#ifdef MVF_MLP_DEBUG
    if (const char *e = getenv("MVF_MLP_WGS")) forced_per_cu = (int)(per_cu = atoi(e));
#endif
    (void)forced_per_cu;
    const i64 want = (rows + MLP_ROWS - 1) / MLP_ROWS, cap = (i64)cus * (per_cu < 1 ? 1 : per_cu > 64 ? 64 : per_cu);
    dim3 grid((unsigned)(want < cap ? want : cap));
    int wmask = 0;
    for (int l = 0; l < nlayers; ++l) wmask |= (layers[l].cout > 32 ? 1 : 0) << l;
    const bool fixed = A.vec_in && A.vec_out && mode == 0;
    // 16-byte pieces of the first layer's row per lane: half of the columns (one k-half)
    const int need = (((cin0 + 1) >> 1) + 3) / 4;
    const int xq = cin0 > MLP_KC ? 8 : need <= 1 ? 1 : need <= 4 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    // The grid is what the device holds at once (registers and LDS of the instantiation
    // decide: looked up once per instantiation and LDS size), every workgroup walking its
    // share of the tiles: a larger grid runs in rounds, the last of them partly empty, and
    // stages the weights once more per extra workgroup.
#define MVF_MLP_RESIDENT(KERNEL_)                                                                \
    do {                                                                                         \
        static size_t lds_seen = ~(size_t)0;                                                     \
        static int occ_seen = 0;                                                                 \
        if (lds_seen != lds) {                                                                   \
            int occ = 0;                                                                         \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, KERNEL_, MLP_THREADS, lds) != hipSuccess || occ < 1) \
                occ = (int)per_cu;                                                               \
            occ_seen = occ;                                                                      \
            lds_seen = lds;                                                                      \
        }                                                                                        \
        i64 cap_ = (i64)cus * occ_seen;                                                          \
        if (forced_per_cu > 0) cap_ = (i64)cus * forced_per_cu;                                  \
        grid = dim3((unsigned)(want < cap_ ? want : cap_));                                      \
    } while (0)
#define MVF_MLP_LAUNCH_X(NL_, WM_, XQ_)                                                          \
    do {                                                                                         \
        MVF_HIP(hipFuncSetAttribute((const void *)mlp_chain_kernel<NL_, WM_, XQ_>,               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        MVF_MLP_RESIDENT((mlp_chain_kernel<NL_, WM_, XQ_>));                                     \
        hipLaunchKernelGGL((mlp_chain_kernel<NL_, WM_, XQ_>), grid, dim3(MLP_THREADS), lds, st, A); \
    } while (0)
#define MVF_MLP_LAUNCH(NL_, WM_) MVF_MLP_LAUNCH_X(NL_, WM_, 8)
    // chains of narrow layers also come with the first layer's row held in 1 or 4
    // 16-byte pieces per lane (8 / 32 input columns) instead of 8
#define MVF_MLP_LAUNCH_NARROW(NL_)                                                               \
    do {                                                                                         \
        if (xq == 1) MVF_MLP_LAUNCH_X(NL_, 0, 1);                                                \
        else if (xq == 4) MVF_MLP_LAUNCH_X(NL_, 0, 4);                                           \
        else MVF_MLP_LAUNCH_X(NL_, 0, 8);                                                        \
    } while (0)
#define MVF_MLP_LAUNCH_POOL_X(NL_, WM_, P_, XQ_)                                                 \
    do {                                                                                         \
        MVF_HIP(hipFuncSetAttribute((const void *)mlp_chain_pool_kernel<NL_, WM_, P_, XQ_>,      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        MVF_MLP_RESIDENT((mlp_chain_pool_kernel<NL_, WM_, P_, XQ_>));                            \
        hipLaunchKernelGGL((mlp_chain_pool_kernel<NL_, WM_, P_, XQ_>), grid, dim3(MLP_THREADS), lds, st, A); \
    } while (0)
#define MVF_MLP_LAUNCH_POOL(NL_, WM_, P_)                                                        \
    do {                                                                                         \
        if ((WM_) == 0 && xq == 1) MVF_MLP_LAUNCH_POOL_X(NL_, WM_, P_, 1);                       \
        else if ((WM_) == 0 && xq == 4) MVF_MLP_LAUNCH_POOL_X(NL_, WM_, P_, 4);                  \
        else MVF_MLP_LAUNCH_POOL_X(NL_, WM_, P_, 8);                                             \
    } while (0)
    // three narrow layers on at most 64 aligned input columns (rows or the dot epilogue)
    // (the output may be of any width and alignment: a score layer of one channel)
    if (g_mlp_precision != 0 && A.vec_in && mode == 0 && nlayers == 3 && wmask == 0 && cin0 <= MLP_KC && !pool_out) {
        const int nq0 = cin0 <= 16 ? 1 : cin0 <= 32 ? 2 : 4;
        const size_t lds3 = bf3_narrow3_lds_bytes(nq0);
#define MVF_BF3N_LAUNCH(NQ_)                                                                     \
    do {                                                                                         \
        MVF_HIP(hipFuncSetAttribute((const void *)mlp_narrow3_bf3_kernel<NQ_>,                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));     \
        static int occ_seen = 0;                                                                 \
        if (!occ_seen && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_seen, mlp_narrow3_bf3_kernel<NQ_>, MLP_THREADS, lds3) != hipSuccess || occ_seen < 1)) \
            occ_seen = 2;                                                                        \
        const i64 cap3 = (i64)cus * occ_seen;                                                    \
        dim3 grid3((unsigned)(want < cap3 ? want : cap3));                                       \
        hipLaunchKernelGGL((mlp_narrow3_bf3_kernel<NQ_>), grid3, dim3(MLP_THREADS), lds3, st, A); \
    } while (0)
        if (nq0 == 1) MVF_BF3N_LAUNCH(1);
        else if (nq0 == 2) MVF_BF3N_LAUNCH(2);
        else MVF_BF3N_LAUNCH(4);
#undef MVF_BF3N_LAUNCH
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    // two wide layers on at most 64 aligned, ungathered input columns: bf16 matrix cores
    if (g_mlp_precision != 0 && fixed && nlayers == 2 && wmask == 3 && cin0 <= MLP_KC && !gather_a && !gather_b && !dot_out &&
        (!pool_out || (layers[1].cout % 4 == 0 && (uintptr_t)pool_out % 16 == 0))) {
        const int nq0 = (cin0 + 15) / 16;
        const size_t lds3 = bf3_lds_bytes(nq0, pool_out != nullptr);
        // two workgroups per CU: the epilogue of one wave runs beside the MFMAs of the other
        int wg_per_cu = 2;
#ifdef MVF_MLP_DEBUG
        if (const char *e = getenv("MVF_BF3_WGS")) wg_per_cu = atoi(e);
#endif
        const i64 cap3 = (i64)cus * (wg_per_cu < 1 ? 1 : wg_per_cu);
        dim3 grid3((unsigned)(want < cap3 ? want : cap3));
        const int pool = !pool_out ? 0 : pool_max ? 3 : pool_groups == 1 ? 1 : 2;
#define MVF_BF3_LAUNCH(NQ_, P_)                                                                  \
    do {                                                                                         \
        MVF_HIP(hipFuncSetAttribute((const void *)mlp_wide2_bf3_kernel<NQ_, P_>,                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));     \
        /* never more workgroups than the device holds at once (the instantiation with the */   \
        /* grouped pool's weights fits one per CU only) */                                      \
        static int occ_seen = 0;                                                                 \
        if (!occ_seen && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_seen, mlp_wide2_bf3_kernel<NQ_, P_>, MLP_THREADS, lds3) != hipSuccess || occ_seen < 1)) \
            occ_seen = 1;                                                                        \
        if (occ_seen < wg_per_cu) grid3 = dim3((unsigned)(want < (i64)cus * occ_seen ? want : (i64)cus * occ_seen)); \
        hipLaunchKernelGGL((mlp_wide2_bf3_kernel<NQ_, P_>), grid3, dim3(MLP_THREADS), lds3, st, A); \
    } while (0)
#define MVF_BF3_LAUNCH_P(NQ_)                                                                    \
    do {                                                                                         \
        if (pool == 0) MVF_BF3_LAUNCH(NQ_, 0);                                                   \
        else if (pool == 1) MVF_BF3_LAUNCH(NQ_, 1);                                              \
        else if (pool == 2) MVF_BF3_LAUNCH(NQ_, 2);                                              \
        else MVF_BF3_LAUNCH(NQ_, 3);                                                             \
    } while (0)
        if (nq0 == 1) MVF_BF3_LAUNCH_P(1);
        else if (nq0 == 2) MVF_BF3_LAUNCH_P(2);
        else if (nq0 == 3) MVF_BF3_LAUNCH_P(3);
        else MVF_BF3_LAUNCH_P(4);
#undef MVF_BF3_LAUNCH
#undef MVF_BF3_LAUNCH_P
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (pool_out) {
        MVF_REQUIRE(fixed && nlayers == 2 && (wmask == 0 || wmask == 3),
                    "mlp_forward_pool: two layers of equal output width on 16-byte aligned, unsummed inputs only");
        if (pool_max) {
            if (wmask == 0) MVF_MLP_LAUNCH_POOL(2, 0, 3);
            else MVF_MLP_LAUNCH_POOL(2, 3, 3);
        } else if (pool_groups == 1) {
            if (wmask == 0) MVF_MLP_LAUNCH_POOL(2, 0, 1);
            else MVF_MLP_LAUNCH_POOL(2, 3, 1);
        } else {
            if (wmask == 0) MVF_MLP_LAUNCH_POOL(2, 0, 2);
            else MVF_MLP_LAUNCH_POOL(2, 3, 2);
        }
    } else if (!fixed) MVF_MLP_LAUNCH(-1, 0);
    else switch (nlayers * 8 + wmask) {
    case 8 + 0: MVF_MLP_LAUNCH_NARROW(1); break;
    case 8 + 1: MVF_MLP_LAUNCH(1, 1); break;
    case 16 + 0: MVF_MLP_LAUNCH_NARROW(2); break;
    case 16 + 1: MVF_MLP_LAUNCH(2, 1); break;
    case 16 + 2: MVF_MLP_LAUNCH(2, 2); break;
    case 16 + 3: MVF_MLP_LAUNCH(2, 3); break;
    case 24 + 0: MVF_MLP_LAUNCH_NARROW(3); break;
    case 24 + 1: MVF_MLP_LAUNCH(3, 1); break;
    case 24 + 2: MVF_MLP_LAUNCH(3, 2); break;
    case 24 + 3: MVF_MLP_LAUNCH(3, 3); break;
    case 24 + 4: MVF_MLP_LAUNCH(3, 4); break;
    case 24 + 5: MVF_MLP_LAUNCH(3, 5); break;
    case 24 + 6: MVF_MLP_LAUNCH(3, 6); break;
    default: MVF_MLP_LAUNCH(3, 7); break;
    }
#undef MVF_MLP_LAUNCH
#undef MVF_MLP_RESIDENT
#undef MVF_MLP_LAUNCH_X
#undef MVF_MLP_LAUNCH_NARROW
#undef MVF_MLP_LAUNCH_POOL
#undef MVF_MLP_LAUNCH_POOL_X
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_mlp_forward(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    float *out, int32_t ldo, void *stream)
{
    return mlp_forward_impl(xa, lda, cols_a, gather_a, xb, ldb, cols_b, gather_b, mode, rows, layers,
                            nlayers, out, ldo, nullptr, 0, nullptr, 0.f, nullptr, stream);
}

extern "C" int mvf_mlp_forward_dot(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    const float *queries, int32_t ldq, const int32_t *row2seg, float divisor, float *dot_out,
    void *stream)
{
    MVF_REQUIRE(dot_out, "mlp_forward_dot: null output");
    return mlp_forward_impl(xa, lda, cols_a, gather_a, xb, ldb, cols_b, gather_b, mode, rows, layers,
                            nlayers, nullptr, 0, queries, ldq, row2seg, divisor, dot_out, stream);
}

// Sums of the segments that cross a tile boundary, from the per-tile partial sums the pool
// epilogue left (MlpArgs::pool_part), added in tile order: the first tile's slot 1, the slots
// 1 of the tiles that lie inside the segment, the last tile's slot 0.  One thread per (tile,
// 4 channels): it completes the segment that begins in its tile and continues behind it, if
// there is one.  Segments inside one tile were stored by the epilogue.
__global__ __launch_bounds__(256) void pool_fixup_kernel(
    const i64 *__restrict__ ptr, const int *__restrict__ row2seg, i64 rows, unsigned tiles, unsigned q,
    const float *__restrict__ part, float *__restrict__ out)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned tile = t / q;
    if (tile >= tiles) return;
    const unsigned cout = 4u * q, c = (t - tile * q) * 4u;
    const i64 first = (i64)tile * POOL_TILE, next = first + POOL_TILE;
    if (next >= rows) return;                          // the last tile: nothing continues behind it
    const int s = row2seg[next - 1];
    const i64 b = ptr[s + 1];
    if (b <= next || ptr[s] < first) return;           // ends here, or began in an earlier tile
    const i64 tb = (b - 1) / POOL_TILE;
    float4 acc = *(const float4 *)(part + ((size_t)tile * 2 + 1) * cout + c);
    for (i64 u = (i64)tile + 1; u <= tb; ++u) {
        const float4 v = *(const float4 *)(part + ((size_t)u * 2 + (u == tb ? 0 : 1)) * cout + c);
        acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y);
        acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
    }
    *(float4 *)(out + (size_t)s * cout + c) = acc;
}

extern "C" size_t mvf_mlp_forward_pool_workspace_bytes(int64_t rows, int32_t cout)
{
    const size_t tiles = (size_t)((rows > 0 ? rows : 1) + POOL_TILE - 1) / POOL_TILE;
    return align_up(tiles * 2 * (size_t)(cout > 0 ? cout : 1) * sizeof(float));
}

extern "C" int mvf_mlp_forward_pool(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    const float *att, int32_t G, const int32_t *row2seg, const int64_t *ptr, int64_t nseg,
    float *pool_out, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(pool_out && ptr && nseg >= 0 && layers && nlayers >= 1, "mlp_forward_pool: null output or pointers");
    const int cout = layers[nlayers - 1].cout;
    MVF_REQUIRE(cout % 4 == 0 && (uintptr_t)pool_out % 16 == 0,
                "mlp_forward_pool: a multiple of 4 output channels and a 16-byte aligned output");
    if (rows == 0 || nseg == 0) return MVF_OK;
    if (!workspace || workspace_bytes < mvf_mlp_forward_pool_workspace_bytes(rows, cout)) {
        set_error("mlp_forward_pool: workspace too small");
        return MVF_E_WORKSPACE;
    }
    if (int rc = mlp_forward_impl(xa, lda, cols_a, gather_a, xb, ldb, cols_b, gather_b, 0, rows, layers,
                                  nlayers, nullptr, 0, nullptr, 0, row2seg, 0.f, nullptr, stream, att, pool_out, G,
                                  false, (float *)workspace))
        return rc;
    const i64 tiles = (rows + POOL_TILE - 1) / POOL_TILE;
    MVF_REQUIRE(tiles * (cout / 4) < (1ll << 32), "mlp_forward_pool: too many rows");
    hipLaunchKernelGGL(pool_fixup_kernel, dim3(cdiv(tiles * (cout / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const i64 *)ptr, row2seg, (i64)rows, (unsigned)tiles, (unsigned)(cout / 4),
                       (const float *)workspace, pool_out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_mlp_forward_segmax(
    const float *xa, int32_t lda, int32_t cols_a, const float *xb, int32_t ldb, int32_t cols_b,
    int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers, float *out, int32_t ldo,
    const int64_t *ptr, int64_t nseg, const int32_t *row2seg, float *seg_out, void *stream)
{
    MVF_REQUIRE(nseg >= 0 && rows >= 0 && layers && nlayers >= 1, "mlp_forward_segmax: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(ptr && seg_out && (uintptr_t)seg_out % 16 == 0 && layers[nlayers - 1].cout % 4 == 0,
                "mlp_forward_segmax: needs the pointers and a 16-byte aligned segment output of a multiple of 4 channels");
    hipLaunchKernelGGL(segmax_prepare_kernel, dim3(cdiv(nseg, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const i64 *)ptr, (i64)nseg, seg_out, layers[nlayers - 1].cout);
    MVF_LAUNCH_CHECK();
    return mlp_forward_impl(xa, lda, cols_a, nullptr, xb, ldb, cols_b, nullptr, 0, rows, layers,
                            nlayers, out, ldo, nullptr, 0, row2seg, 0.f, nullptr, stream, nullptr, seg_out, 1, true);
}

extern "C" int mvf_attention_pool(
    const float *keys, int32_t ldk, const float *queries, int32_t ldq, const int32_t *row2seg,
    const float *x_mod, int32_t C, const int64_t *ptr, int64_t nseg, int64_t rows, int32_t G,
    int32_t D, int32_t dim_scaling, int32_t group_scaling, float eps, const int32_t *chan2group,
    const float *gate_w, const float *gate_b, int32_t gating,
    float *comp, float *att, float *gate, float *out, void *stream)
{
    MVF_REQUIRE(nseg >= 0 && rows >= 0 && G >= 1 && C >= 1, "attention_pool: bad shape");
    MVF_REQUIRE(G == 1 || chan2group, "attention_pool: groups need the channel map");
    MVF_REQUIRE(!keys || (queries && D >= 1 && ldk >= G * D && ldq >= G * D),
                "attention_pool: bad key / query layout");
    MVF_REQUIRE(!gating || gate, "attention_pool: gating needs the gate buffer");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(ptr && out && (rows == 0 || (comp && att)), "attention_pool: null buffer");
    // x_mod == NULL: scores only; `out` is prepared for mvf_mlp_forward_pool
    const bool prepare = x_mod == nullptr && rows > 0;
    MVF_REQUIRE(!prepare || (G <= POOL_MAX_GROUPS && C % G == 0 && (C / G) % 4 == 0 && (uintptr_t)out % 16 == 0),
                "attention_pool: the fused pool needs at most 8 equal groups of a multiple of 4 channels and a 16-byte aligned output");
    hipStream_t st = (hipStream_t)stream;
    if (keys && rows > 0) {
        const float dim_div = dim_scaling ? (float)sqrt((double)D) : 0.f;
        hipLaunchKernelGGL(attention_comp_kernel, dim3(cdiv(rows * G, 256)), dim3(256), 0, st,
                           keys, ldk, queries, ldq, row2seg, (i64)rows, G, D, dim_div, comp);
    }
    hipLaunchKernelGGL(attention_scores_kernel, dim3(cdiv(nseg * G, 256)), dim3(256), 0, st,
                       (const i64 *)ptr, (i64)nseg, G, group_scaling, eps, gate_w, gate_b,
                       (const float *)comp, att, gating ? gate : (float *)nullptr,
                       prepare ? out : (float *)nullptr, C);
    if (prepare) {
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    // 16-byte vectors when no vector of 4 channels straddles two groups
    const bool vec4 = C % 4 == 0 && ((uintptr_t)x_mod % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                      (G == 1 || (C % G == 0 && (C / G) % 4 == 0));
    if (vec4)
        hipLaunchKernelGGL((attention_pool_kernel<4>), dim3(cdiv(nseg * (C / 4), 256 * ATT_ILP)), dim3(256), 0, st,
                           x_mod, C, att, G, G == 1 ? (const int *)nullptr : chan2group,
                           gating ? gate : (const float *)nullptr, (const i64 *)ptr, (i64)nseg, out);
    else
        hipLaunchKernelGGL((attention_pool_kernel<1>), dim3(cdiv(nseg * C, 256 * ATT_ILP)), dim3(256), 0, st,
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
    int64_t nseg, int32_t pick_max, float *out, int64_t *picked, void *stream)
{
    MVF_REQUIRE(nseg >= 0 && C >= 1 && ldf >= 1, "segment_pick: bad shape");
    if (nseg == 0) return MVF_OK;
    MVF_REQUIRE(x_mod && feat && ptr && out, "segment_pick: null buffer");
    hipLaunchKernelGGL(segment_pick_kernel, dim3(cdiv(nseg * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x_mod, C, feat, ldf, (const i64 *)ptr, (i64)nseg,
                       pick_max, out, (i64 *)picked);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
