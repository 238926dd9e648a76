// Neighbourhood-based mapping features (core/data_transform/multimodal/
// image.py:431-612): exact k nearest neighbours of every point among the
// points of the cloud, and the occlusion ratio of every (point, view) mapping.
// gfx950 only.
//
// k-NN: brute force, 256 queries per block, one query per thread with its K
// best (distance, index) pairs kept sorted in registers; the candidate is the
// same for all lanes and arrives through scalar loads.  Distances are the float64 sums dx*dx + dy*dy + dz*dz of the
// float32 coordinates (differences and squares are exact in float64), so the
// ranking is that of exact arithmetic up to the last rounding of the sum, and
// equal distances rank by ascending index: the result does not depend on the
// traversal order.  The point itself is part of the candidates, as in the
// reference (its own distance 0 comes first).
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

// The candidates are cut into gridDim.y slices so that small clouds still fill
// the chip; slice s of query q leaves its K best pairs, sorted, in
// part_d / part_i [q][s][K] (written straight to neighbors when there is one
// slice only).
template <int K>
__global__ __launch_bounds__(256) void knn_kernel(
    const float *__restrict__ xyz, i64 n, int k, i64 slice_len, double *__restrict__ part_d,
    int *__restrict__ part_i, i64 *__restrict__ neighbors)
{
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < n;
    const i64 qs = live ? q : n - 1;
    const double qx = (double)xyz[3 * qs], qy = (double)xyz[3 * qs + 1], qz = (double)xyz[3 * qs + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = 1.0e300; bi[t] = 0x7fffffff; }
    const i64 lo = (i64)blockIdx.y * slice_len;
    const i64 hi = lo + slice_len < n ? lo + slice_len : n;
    // the candidate is the same for every lane: its coordinates come through
    // scalar loads, eight candidates requested at a time
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
            const int j = (int)(base + u);
            if (base + u < hi && knn_before(d, j, bd[K - 1], bi[K - 1])) knn_insert<K>(bd, bi, d, j);
        }
    }
    if (!live) return;
    if (gridDim.y == 1) {
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < k) neighbors[(size_t)q * k + t] = (i64)bi[t];
        return;
    }
    const size_t o = ((size_t)q * gridDim.y + blockIdx.y) * K;
#pragma unroll
    for (int t = 0; t < K; ++t) { part_d[o + t] = bd[t]; part_i[o + t] = bi[t]; }
}

// Merge of the slices' sorted lists of one query: one thread per query.
template <int K>
__global__ __launch_bounds__(256) void knn_merge_kernel(
    const double *__restrict__ part_d, const int *__restrict__ part_i, i64 n, int k, int slices,
    i64 *__restrict__ neighbors)
{
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = 1.0e300; bi[t] = 0x7fffffff; }
    for (int s = 0; s < slices; ++s) {
        const size_t o = ((size_t)q * slices + s) * K;
        for (int t = 0; t < K; ++t) {
            const double d = part_d[o + t];
            const int j = part_i[o + t];
            if (!knn_before(d, j, bd[K - 1], bi[K - 1])) break;     // the list is sorted
            knn_insert<K>(bd, bi, d, j);
        }
    }
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

// Slices of the candidate range: enough blocks for about eight waves per SIMD,
// at least 4096 candidates per slice.
static int knn_slices(i64 n)
{
    const i64 blocks = (n + 255) / 256;
    i64 s = (512 + blocks - 1) / blocks;
    const i64 most = (n + 4095) / 4096;
    if (s > most) s = most;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

extern "C" size_t mvf_knn_workspace_bytes(int64_t n, int32_t k)
{
    const int slices = knn_slices(n);
    if (slices == 1) return 256;
    const size_t e = (size_t)n * slices * knn_width(k);
    return align_up(e * sizeof(double)) + align_up(e * sizeof(int));
}

extern "C" int mvf_knn(const float *xyz, int64_t n, int32_t k, int64_t *neighbors, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(n >= 1 && n < (1ll << 31) && k >= 1 && k <= 64 && k <= n,
                "knn: need 1 <= k <= min(64, n), got n=%lld k=%d", (long long)n, k);
    MVF_REQUIRE(xyz && neighbors, "knn: null pointer");
    const int slices = knn_slices(n), K = knn_width(k);
    const i64 slice_len = (n + slices - 1) / slices;
    Arena ar(workspace, workspace_bytes);
    double *pd = nullptr;
    int *pi = nullptr;
    if (slices > 1) {
        pd = ar.take<double>((size_t)n * slices * K);
        pi = ar.take<int>((size_t)n * slices * K);
        if (!ar.ok()) { set_error("knn: workspace too small (%zu < %zu)", workspace_bytes, ar.off); return MVF_E_WORKSPACE; }
    }
    dim3 grid(cdiv(n, 256), slices), block(256), mgrid(cdiv(n, 256));
    hipStream_t st = (hipStream_t)stream;
#define MVF_KNN(K_)                                                                                  \
    do {                                                                                             \
        hipLaunchKernelGGL((knn_kernel<K_>), grid, block, 0, st, xyz, (i64)n, k, slice_len, pd, pi,   \
                           (i64 *)neighbors);                                                        \
        if (slices > 1)                                                                              \
            hipLaunchKernelGGL((knn_merge_kernel<K_>), mgrid, block, 0, st, pd, pi, (i64)n, k, slices, \
                               (i64 *)neighbors);                                                    \
    } while (0)
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
