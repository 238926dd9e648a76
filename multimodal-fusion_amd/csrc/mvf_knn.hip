// Neighbourhood-based mapping features (core/data_transform/multimodal/
// image.py:431-612): exact k nearest neighbours of every point among the
// points of the cloud, and the occlusion ratio of every (point, view) mapping.
// gfx950 only.
//
// k-NN: brute force, 256 queries per block, one query per thread with its K
// best (distance, index) pairs kept sorted in registers; the candidate is the
// same for all lanes and arrives through scalar loads.  Distances are the
// float64 sums dx*dx + dy*dy + dz*dz of the float32 coordinates (differences
// and squares are exact in float64), so the ranking is that of exact
// arithmetic up to the last rounding of the sum, and equal distances rank by
// ascending point id: the result does not depend on the traversal order.  The
// point itself is part of the candidates, as in the reference (its own
// distance 0 comes first).
//
// The expensive part is the sorted insertion, which a wave executes whenever
// one of its lanes accepts a candidate.  The caller therefore passes the points
// in a space-filling-curve order (ids = original index of each row): the
// queries of a block are neighbours in space, the block scans the candidates
// next to it in that order first, and after this window the lists are close to
// final, so that the scan of the other candidates rarely inserts -- and most
// of them are not scanned at all: see the chunk boxes of knn_kernel.
#include "mvf_common.h"

namespace mvf {

__device__ __forceinline__ bool knn_before(double d, int j, double e, int i)
{
    return d < e || (d == e && j < i);
}

// Sorted insertion of (d, j) into the K best pairs of a thread (registers).
template <int K>
__device__ __forceinline__ void knn_insert(double (&bd)[K], int (&bi)[K], double d, int j)
{
    bool placed = false;
#pragma unroll
    for (int t = K - 1; t >= 1; --t) {
        const bool mv = knn_before(d, j, bd[t - 1], bi[t - 1]);
        if (mv) { bd[t] = bd[t - 1]; bi[t] = bi[t - 1]; }
        else if (!placed) { bd[t] = d; bi[t] = j; placed = true; }
    }
    if (!placed) { bd[0] = d; bi[0] = j; }
}

constexpr int KNN_WINDOW = 2048;      // candidates scanned first, on each side of the block

template <int K>
__device__ __forceinline__ void knn_scan(
    const float *__restrict__ xyz, const i64 *__restrict__ ids, i64 lo, i64 hi, double qx, double qy,
    double qz, double (&bd)[K], int (&bi)[K])
{
    for (i64 base = lo; base < hi; base += 8) {
        double cx[8], cy[8], cz[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const i64 j = base + u < hi ? base + u : hi - 1;
            cx[u] = (double)xyz[3 * j]; cy[u] = (double)xyz[3 * j + 1]; cz[u] = (double)xyz[3 * j + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double dx = qx - cx[u], dy = qy - cy[u], dz = qz - cz[u];
            const double d = dx * dx + dy * dy + dz * dz;
            if (base + u < hi && d <= bd[K - 1]) {
                const int j = ids ? (int)ids[base + u] : (int)(base + u);
                if (knn_before(d, j, bd[K - 1], bi[K - 1])) knn_insert<K>(bd, bi, d, j);
            }
        }
    }
}

// Rows come in chunks of 256 (= the queries of a block) with bounding boxes
// box_lo / box_hi [chunks][3] (NULL: no culling).  After the window the block
// knows the largest k-th distance of its queries; a chunk whose box is farther
// than that from the box of the queries cannot change any list and is skipped.
template <int K>
__global__ __launch_bounds__(256) void knn_kernel(
    const float *__restrict__ xyz, const i64 *__restrict__ ids, i64 n, int k,
    const float *__restrict__ box_lo, const float *__restrict__ box_hi,
    i64 *__restrict__ neighbors)
{
    __shared__ unsigned long long s_worst;
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < n;
    const i64 qs = live ? q : n - 1;
    const double qx = (double)xyz[3 * qs], qy = (double)xyz[3 * qs + 1], qz = (double)xyz[3 * qs + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = 1.0e300; bi[t] = 0x7fffffff; }
    if (threadIdx.x == 0) s_worst = 0ull;
    const i64 chunks = (n + 255) / 256;
    const i64 c0 = blockIdx.x;
    const i64 cw_lo = c0 > KNN_WINDOW / 256 ? c0 - KNN_WINDOW / 256 : 0;
    const i64 cw_hi = c0 + 1 + KNN_WINDOW / 256 < chunks ? c0 + 1 + KNN_WINDOW / 256 : chunks;
    const i64 w_lo = cw_lo * 256, w_hi = cw_hi * 256 < n ? cw_hi * 256 : n;
    knn_scan<K>(xyz, ids, w_lo, w_hi, qx, qy, qz, bd, bi);
    if (box_lo == nullptr) {
        knn_scan<K>(xyz, ids, 0, w_lo, qx, qy, qz, bd, bi);
        knn_scan<K>(xyz, ids, w_hi, n, qx, qy, qz, bd, bi);
    } else {
        // positive doubles order like their bit patterns
        __syncthreads();
        atomicMax(&s_worst, (unsigned long long)__double_as_longlong(bd[K - 1]));
        __syncthreads();
        const double worst = __longlong_as_double((long long)s_worst);
        double ql[3], qh[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { ql[a] = (double)box_lo[3 * c0 + a]; qh[a] = (double)box_hi[3 * c0 + a]; }
        for (i64 c = 0; c < chunks; ++c) {
            if (c >= cw_lo && c < cw_hi) continue;
            double gap2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)box_lo[3 * c + a], hi = (double)box_hi[3 * c + a];
                double g = lo - qh[a] > ql[a] - hi ? lo - qh[a] : ql[a] - hi;
                g = g > 0.0 ? g : 0.0;
                gap2 += g * g;
            }
            if (gap2 * (1.0 - 1e-12) > worst) continue;
            const i64 lo_r = c * 256, hi_r = lo_r + 256 < n ? lo_r + 256 : n;
            knn_scan<K>(xyz, ids, lo_r, hi_r, qx, qy, qz, bd, bi);
        }
    }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < K; ++t)
        if (t < k) neighbors[(size_t)q * k + t] = (i64)bi[t];
}

// Occlusion of a (point, view) mapping: share of the point's k nearest
// neighbours (the point itself counted once more, as image.py:587-596 does)
// that are mapped to the same image.  seen: [n_points][n_images] bytes.
// One thread per view-level row.
__global__ __launch_bounds__(256) void occlusion_kernel(
    const i64 *__restrict__ neighbors, int k_stride, int k, const i64 *__restrict__ row_point,
    const i64 *__restrict__ images, i64 rows, const uint8_t *__restrict__ seen, i64 n_images,
    float *__restrict__ out, int out_stride)
{
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const i64 p = row_point[r], im = images[r];
    float cnt = 1.0f;
    for (int j = 0; j < k; ++j)
        cnt += (float)seen[(size_t)neighbors[(size_t)p * k_stride + j] * n_images + im];
    out[(size_t)r * out_stride] = __fdiv_rn(cnt, (float)(k + 1));
}

}  // namespace mvf

using namespace mvf;

static inline int knn_width(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 24 ? 24 : k <= 32 ? 32 : k <= 48 ? 48 : 64; }

extern "C" int mvf_knn(const float *xyz, const int64_t *ids, int64_t n, int32_t k,
                       const float *box_lo, const float *box_hi, int64_t *neighbors, void *stream)
{
    MVF_REQUIRE(n >= 1 && n < (1ll << 31) && k >= 1 && k <= 64 && k <= n,
                "knn: need 1 <= k <= min(64, n), got n=%lld k=%d", (long long)n, k);
    MVF_REQUIRE(xyz && neighbors, "knn: null pointer");
    MVF_REQUIRE((box_lo == nullptr) == (box_hi == nullptr), "knn: box_lo and box_hi go together");
    const int K = knn_width(k);
    dim3 grid(cdiv(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MVF_KNN(K_) hipLaunchKernelGGL((knn_kernel<K_>), grid, block, 0, st, xyz, (const i64 *)ids, (i64)n, k, \
                                       box_lo, box_hi, (i64 *)neighbors)
    if (K == 8) MVF_KNN(8); else if (K == 16) MVF_KNN(16); else if (K == 24) MVF_KNN(24);
    else if (K == 32) MVF_KNN(32); else if (K == 48) MVF_KNN(48); else MVF_KNN(64);
#undef MVF_KNN
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_mapping_occlusion(
    const int64_t *neighbors, int32_t k_stride, int32_t k, const int64_t *row_point,
    const int64_t *images, int64_t rows, const uint8_t *seen, int64_t n_images, float *out,
    int32_t out_stride, void *stream)
{
    MVF_REQUIRE(k >= 1 && k <= k_stride && rows >= 0 && n_images >= 1 && out_stride >= 1,
                "mapping_occlusion: bad arguments");
    if (rows == 0) return MVF_OK;
    MVF_REQUIRE(neighbors && row_point && images && seen && out, "mapping_occlusion: null pointer");
    hipLaunchKernelGGL(occlusion_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const i64 *)neighbors, k_stride, k, (const i64 *)row_point, (const i64 *)images,
                       (i64)rows, seen, (i64)n_images, out, out_stride);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
