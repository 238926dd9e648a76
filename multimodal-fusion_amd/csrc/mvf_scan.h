// Device-wide exclusive scan and order-preserving compaction helpers.
// Wavefront = 64 lanes: flags are ranked with __ballot + popcount inside a
// wave, across the 4 waves of a 256-thread block through LDS, and across
// blocks with a small scan over per-block counts, so outputs keep input order.
#pragma once

#include "mvf_common.h"

namespace mvf {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread, blocked
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // elements per block

// Exclusive rank of `flag` among the 256 threads of the block, block total in `total`.
__device__ __forceinline__ unsigned block_rank_flag(bool flag, unsigned &total)
{
    __shared__ unsigned wave_cnt[4];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 m = __ballot(flag);
    const unsigned below = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    unsigned off = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < 4; ++w) {
        const unsigned c = wave_cnt[w];
        if (w < wave) off += c;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return off + below;
}

// Block-wide exclusive scan of one value per thread (NW waves of 64 threads).
template <typename T, int NW = 4>
__device__ __forceinline__ T block_exclusive_scan(T v, T &total)
{
    __shared__ T wave_sum[NW];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < NW; ++w) {
        const T c = wave_sum[w];
        if (w < wave) off += c;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return off + inc - v;
}

// ---- single-launch scan for small arrays (tile counters, block sums) --------
// One 1024-thread block; thread t owns the contiguous chunk [t*per, (t+1)*per).
// out[i] = sum(in[0..i)), out[n] = total; in == out is allowed.
template <typename TI, typename TO, typename TT>
__global__ __launch_bounds__(1024) void scan_single_block_kernel(
    const TI *in, i64 n, TO *out, TT *__restrict__ total_out)
{
    const i64 per = (n + 1023) / 1024;
    const i64 lo = (i64)threadIdx.x * per;
    const i64 hi = lo + per < n ? lo + per : n;
    TO s = 0;
    for (i64 i = lo; i < hi; ++i) s += (TO)in[i];
    TO total;
    TO run = block_exclusive_scan<TO, 16>(s, total);
    for (i64 i = lo; i < hi; ++i) {
        const TO v = (TO)in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        out[n] = total;
        if (total_out) *total_out = (TT)total;
    }
}

// ---- 3-kernel exclusive scan: out[i] = sum(in[0..i)), out[n] = total --------
template <typename TI, typename TO>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(
    const TI *__restrict__ in, i64 n, TO *__restrict__ block_sums)
{
    const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
    TO s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += (TO)in[base + k];
    TO total;
    block_exclusive_scan<TO>(s, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: in-place exclusive scan of block_sums[0..nb), grand total to
// *total_out (TT) when not null.
template <typename TO, typename TT>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_spine_kernel(
    TO *__restrict__ block_sums, i64 nb, TT *__restrict__ total_out)
{
    TO carry = 0;
    for (i64 base = 0; base < nb; base += SCAN_BLOCK) {
        const i64 i = base + threadIdx.x;
        const TO v = i < nb ? block_sums[i] : (TO)0;
        TO total;
        const TO ex = block_exclusive_scan<TO>(v, total);
        if (i < nb) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = (TT)carry;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(
    const TI *__restrict__ in, i64 n, const TO *__restrict__ block_sums,
    TO *__restrict__ out)
{
    const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
    TO v[SCAN_ITEMS];
    TO s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? (TO)in[base + k] : (TO)0;
        s += v[k];
    }
    TO total;
    TO run = block_exclusive_scan<TO>(s, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    // the element one past the end receives the grand total
    if (base <= n - 1 && n - 1 < base + SCAN_ITEMS) out[n] = run;
}

static inline size_t scan_workspace_elems(i64 n) { return (size_t)cdiv(n > 0 ? n : 1, SCAN_TILE) + 1; }

// out must hold n + 1 elements.  block_sums: scan_workspace_elems(n) of TO.
template <typename TI, typename TO, typename TT>
static inline void exclusive_scan(
    const TI *in, i64 n, TO *out, TO *block_sums, TT *total_out, hipStream_t st)
{
    const unsigned nb = cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL((scan_reduce_kernel<TI, TO>), dim3(nb), dim3(SCAN_BLOCK), 0, st, in, n, block_sums);
    hipLaunchKernelGGL((scan_spine_kernel<TO, TT>), dim3(1), dim3(SCAN_BLOCK), 0, st, block_sums, (i64)nb, total_out);
    hipLaunchKernelGGL((scan_apply_kernel<TI, TO>), dim3(nb), dim3(SCAN_BLOCK), 0, st, in, n, block_sums, out);
}

// ---- order-preserving compaction: one element per thread -------------------
// Pred: __device__ bool operator()(i64 i) const.
// Emit: __device__ void operator()(i64 i, i64 pos) const.
template <class Pred>
__global__ __launch_bounds__(256) void compact_count_kernel(Pred pred, i64 n, unsigned *__restrict__ block_counts)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && pred(i);
    unsigned total;
    block_rank_flag(f, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

template <class Pred, class Emit>
__global__ __launch_bounds__(256) void compact_emit_kernel(
    Pred pred, Emit emit, i64 n, const unsigned *__restrict__ block_offsets, i64 capacity,
    i64 *__restrict__ overflow_flag)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && pred(i);
    unsigned total;
    const unsigned r = block_rank_flag(f, total);
    if (f) {
        const i64 pos = (i64)block_offsets[blockIdx.x] + r;
        if (pos < capacity) emit(i, pos);
        else if (overflow_flag) *overflow_flag = 1;
    }
}

static inline size_t compact_workspace_elems(i64 n) { return (size_t)cdiv(n > 0 ? n : 1, 256) + 1; }

// block_counts: compact_workspace_elems(n) unsigned.  count_out <- number selected.
template <class Pred, class Emit>
static inline void compact(
    Pred pred, Emit emit, i64 n, unsigned *block_counts, i64 *count_out,
    i64 capacity, i64 *overflow_flag, hipStream_t st)
{
    const unsigned nb = cdiv(n, 256);
    hipLaunchKernelGGL((compact_count_kernel<Pred>), dim3(nb), dim3(256), 0, st, pred, n, block_counts);
    hipLaunchKernelGGL((scan_spine_kernel<unsigned, i64>), dim3(1), dim3(SCAN_BLOCK), 0, st, block_counts, (i64)nb, count_out);
    hipLaunchKernelGGL((compact_emit_kernel<Pred, Emit>), dim3(nb), dim3(256), 0, st, pred, emit, n, block_counts, capacity, overflow_flag);
}

}  // namespace mvf
