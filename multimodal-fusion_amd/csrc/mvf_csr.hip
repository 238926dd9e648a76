// Nested-CSR assembly from dense rows on the device: ImageMapping.from_dense
// (core/multimodal/image.py:1738-1805) and the grouped feature mean of
// ImageMapping.select_points(mode='merge') (:2177-2287).  gfx950 only.
//
// The reference sorts composite int64 keys on the host (np.argsort / np.unique)
// or with torch.sort / torch.unique on the GPU, several times per call, with a
// repeat_interleave temporary per level.  Here one 64-bit key per row,
//   (point * num_images + image) << 32 | x << 16 | y,
// goes through one stable radix sort (rocPRIM's device-wide sort is the one
// library primitive used; only the significant bits are sorted), and flags,
// one scan, and two fill kernels build every level of the CSR from the sorted
// keys: pixels of a view in (x, y) order, views of a point in image order,
// duplicates dropped when asked, features averaged per view in sorted order.
#include <cstring>

#include "mvf_common.h"
#include "mvf_scan.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace mvf {

__global__ __launch_bounds__(256) void fd_keys_kernel(
    const i64 *__restrict__ point_ids, const i64 *__restrict__ image_ids, const int *__restrict__ row2view,
    const int16_t *__restrict__ pixels, i64 rows, i64 num_images, u64 *__restrict__ keys,
    unsigned *__restrict__ vals)
{
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const i64 src = row2view ? (i64)row2view[r] : r;
    const u64 group = (u64)point_ids[src] * (u64)num_images + (u64)image_ids[src];
    keys[r] = (group << 32) | ((u64)(uint16_t)pixels[2 * r] << 16) | (u64)(uint16_t)pixels[2 * r + 1];
    vals[r] = (unsigned)r;
}

// flags[s] = (row s of the sorted order is kept) | (it opens a view) << 32
__global__ __launch_bounds__(256) void fd_flags_kernel(
    const u64 *__restrict__ keys, i64 rows, int dedupe, u64 *__restrict__ flags)
{
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s >= rows) return;
    const u64 k = keys[s];
    const bool first = s == 0;
    const u64 prev = first ? 0 : keys[s - 1];
    const u64 kept = (!dedupe || first || k != prev) ? 1ull : 0ull;
    const u64 opens = (first || (k >> 32) != (prev >> 32)) ? 1ull : 0ull;
    flags[s] = kept | (opens << 32);
}

// pos[s] = exclusive scan of flags: kept rows before s (low word), views opened
// before s (high word); pos[rows] = the totals.
__global__ __launch_bounds__(256) void fd_fill_kernel(
    const u64 *__restrict__ keys, const unsigned *__restrict__ vals, const u64 *__restrict__ flags,
    const u64 *__restrict__ pos, const int16_t *__restrict__ pixels, i64 rows, i64 num_images,
    i64 *__restrict__ images, i64 *__restrict__ view_pointers, int16_t *__restrict__ out_pixels,
    unsigned *__restrict__ view_start, unsigned *__restrict__ point_views, i64 *__restrict__ counters)
{
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s > rows) return;
    if (s == rows) {
        const u64 tot = pos[rows];
        const i64 n_views = (i64)(tot >> 32), n_pix = (i64)(tot & 0xffffffffu);
        view_pointers[n_views] = n_pix;
        view_start[n_views] = (unsigned)rows;
        counters[0] = n_views;
        counters[1] = n_pix;
        return;
    }
    const u64 f = flags[s], p = pos[s];
    const i64 pix = (i64)(p & 0xffffffffu), view = (i64)(p >> 32);
    if (f & 1ull) {
        const unsigned r = vals[s];
        *(int *)(out_pixels + 2 * pix) = *(const int *)(pixels + 2 * (size_t)r);
    }
    if (f >> 32) {
        const u64 group = keys[s] >> 32;
        images[view] = (i64)(group % (u64)num_images);
        view_pointers[view] = pix;
        view_start[view] = (unsigned)s;
        atomicAdd(&point_views[group / (u64)num_images], 1u);
    }
}

// Mean of the kept rows' features per view, rows added in sorted order.
__global__ __launch_bounds__(256) void fd_features_kernel(
    const unsigned *__restrict__ vals, const u64 *__restrict__ flags, const unsigned *__restrict__ view_start,
    const int *__restrict__ row2view, const float *__restrict__ features, int F, const i64 *__restrict__ counters,
    float *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 n_views = counters[0];
    if (e >= n_views * F) return;
    const i64 v = e / F;
    const int c = (int)(e - v * F);
    float sum = 0.f;
    unsigned n = 0;
    for (unsigned s = view_start[v]; s < view_start[v + 1]; ++s) {
        if (!(flags[s] & 1ull)) continue;
        const unsigned r = vals[s];
        const size_t src = row2view ? (size_t)row2view[r] : (size_t)r;
        sum = __fadd_rn(sum, features[src * F + c]);
        ++n;
    }
    out[e] = __fdiv_rn(sum, (float)(n ? n : 1u));
}

// ---- grouped mean: out[r] = mean of the rows that share r's key -------------
__global__ __launch_bounds__(256) void gm_keys_kernel(
    const i64 *__restrict__ keys_in, i64 n, u64 *__restrict__ keys, unsigned *__restrict__ vals)
{
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    keys[r] = (u64)keys_in[r];
    vals[r] = (unsigned)r;
}

// One thread per group (its first sorted row) and channel: rows added in their
// input order (the sort is stable), the mean stored to every row of the group.
__global__ __launch_bounds__(256) void gm_mean_kernel(
    const u64 *__restrict__ keys, const unsigned *__restrict__ vals, i64 n, const float *__restrict__ feats,
    int F, float *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * F) return;
    const i64 s = e / F;
    const int c = (int)(e - s * F);
    const u64 k = keys[s];
    if (s > 0 && keys[s - 1] == k) return;
    float sum = 0.f;
    i64 t = s;
    for (; t < n && keys[t] == k; ++t) sum = __fadd_rn(sum, feats[(size_t)vals[t] * F + c]);
    const float mean = __fdiv_rn(sum, (float)(t - s));
    for (i64 u = s; u < t; ++u) out[(size_t)vals[u] * F + c] = mean;
}

// ---- per-image statistics of a mapping: pixel count and bounding box --------
// stats[i] = {pixels, x min, x max, y min, y max} of image i.  A block walks a
// contiguous chunk of views and keeps the partial results of up to IS_SLOTS
// images in LDS, so that the global atomics are one set per (block, image)
// instead of one per view.
constexpr int IS_SLOTS = 1024;
constexpr int IS_VIEWS_PER_BLOCK = 256 * 16;
constexpr i64 IS_MAX = 0x7fffffffffffffffll, IS_MIN = -IS_MAX - 1;

__global__ __launch_bounds__(256) void image_stats_init_kernel(i64 *__restrict__ stats, i64 n_img)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_img * 5) return;
    const int c = (int)(i % 5);
    stats[i] = c == 0 ? 0 : ((c & 1) ? IS_MAX : IS_MIN);
}

__device__ __forceinline__ void image_stats_merge(i64 *slot, i64 n, i64 x0, i64 x1, i64 y0, i64 y1)
{
    atomicAdd((unsigned long long *)slot, (unsigned long long)n);
    atomicMin((long long *)slot + 1, (long long)x0);
    atomicMax((long long *)slot + 2, (long long)x1);
    atomicMin((long long *)slot + 3, (long long)y0);
    atomicMax((long long *)slot + 4, (long long)y1);
}

template <typename PIX>
__global__ __launch_bounds__(256) void image_stats_kernel(
    const i64 *__restrict__ view_ptr, const i64 *__restrict__ images, const PIX *__restrict__ pixels,
    i64 n_views, i64 n_img, i64 *__restrict__ stats)
{
    __shared__ i64 part[IS_SLOTS * 5];
    const bool staged = n_img <= IS_SLOTS;
    if (staged) {
        for (int i = threadIdx.x; i < (int)n_img * 5; i += 256) {
            const int c = i % 5;
            part[i] = c == 0 ? 0 : ((c & 1) ? IS_MAX : IS_MIN);
        }
        __syncthreads();
    }
    const i64 first = (i64)blockIdx.x * IS_VIEWS_PER_BLOCK;
    const i64 last = first + IS_VIEWS_PER_BLOCK < n_views ? first + IS_VIEWS_PER_BLOCK : n_views;
    for (i64 v = first + threadIdx.x; v < last; v += 256) {
        const i64 b = view_ptr[v], e = view_ptr[v + 1];
        const i64 img = images[v];
        if (e <= b || img < 0 || img >= n_img) continue;
        i64 x0 = IS_MAX, x1 = IS_MIN, y0 = IS_MAX, y1 = IS_MIN;
        for (i64 p = b; p < e; ++p) {
            const i64 x = (i64)pixels[2 * p], y = (i64)pixels[2 * p + 1];
            x0 = x < x0 ? x : x0;
            x1 = x > x1 ? x : x1;
            y0 = y < y0 ? y : y0;
            y1 = y > y1 ? y : y1;
        }
        image_stats_merge(staged ? part + img * 5 : stats + img * 5, e - b, x0, x1, y0, y1);
    }
    if (!staged) return;
    __syncthreads();
    for (int i = threadIdx.x; i < (int)n_img; i += 256) {
        const i64 *q = part + i * 5;
        if (q[0] > 0) image_stats_merge(stats + (i64)i * 5, q[0], q[1], q[2], q[3], q[4]);
    }
}

// Images without a pixel report an all-zero row.
__global__ __launch_bounds__(256) void image_stats_finish_kernel(i64 *__restrict__ stats, i64 n_img)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_img) return;
    if (stats[i * 5] == 0) {
        for (int c = 1; c < 5; ++c) stats[i * 5 + c] = 0;
    }
}

static int sort_bits(u64 max_value)
{
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

// Buffers of both entry points inside the caller's workspace.
struct SortWorkspace {
    u64 *keys_a, *keys_b, *flags, *pos, *scan_blocks;
    unsigned *vals_a, *vals_b, *view_start, *point_views;
    i64 *ptr_blocks;
    void *sort_tmp;
    size_t sort_tmp_bytes, used;
    bool carve(void *workspace, size_t workspace_bytes, i64 rows, i64 num_points)
    {
        Arena ar(workspace, workspace_bytes);
        const size_t n = (size_t)(rows > 0 ? rows : 1);
        keys_a = ar.take<u64>(n);
        keys_b = ar.take<u64>(n);
        vals_a = ar.take<unsigned>(n);
        vals_b = ar.take<unsigned>(n);
        flags = ar.take<u64>(n + 1);
        pos = ar.take<u64>(n + 1);
        scan_blocks = ar.take<u64>(scan_workspace_elems((i64)n + 1));
        view_start = ar.take<unsigned>(n + 1);
        point_views = ar.take<unsigned>((size_t)(num_points > 0 ? num_points : 1) + 1);
        ptr_blocks = ar.take<i64>(scan_workspace_elems(num_points > 0 ? num_points : 1));
        sort_tmp_bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, sort_tmp_bytes, (const u64 *)nullptr, (u64 *)nullptr,
                                        (const unsigned *)nullptr, (unsigned *)nullptr, n, 0u, 64u,
                                        (hipStream_t)0);
        sort_tmp = ar.take<char>(sort_tmp_bytes);
        used = ar.off;
        return ar.ok();
    }
};

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_mapping_from_dense_workspace_bytes(int64_t rows, int64_t num_points)
{
    SortWorkspace ws;
    ws.carve(nullptr, 0, rows, num_points);
    return ws.used;
}

extern "C" int mvf_mapping_from_dense(
    const int64_t *point_ids, const int64_t *image_ids, const int32_t *row2view, const int16_t *pixels,
    const float *features, int32_t F, int64_t rows, int64_t num_points, int64_t num_images,
    int32_t dedupe, int64_t *pointers, int64_t *images, int64_t *view_pointers, int16_t *out_pixels,
    float *out_features, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(rows >= 0 && rows < (1ll << 32) - 1 && num_points >= 1 && num_images >= 1 && F >= 0,
                "mapping_from_dense: bad sizes rows=%lld num_points=%lld num_images=%lld", (long long)rows,
                (long long)num_points, (long long)num_images);
    MVF_REQUIRE((double)num_points * (double)num_images < 4294967296.0,
                "mapping_from_dense: num_points * num_images = %lld * %lld does not fit 32 bits",
                (long long)num_points, (long long)num_images);
    MVF_REQUIRE(pointers && counters, "mapping_from_dense: null output");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        MVF_HIP(hipMemsetAsync(pointers, 0, (size_t)(num_points + 1) * sizeof(int64_t), st));
        MVF_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(int64_t), st));
        if (view_pointers) MVF_HIP(hipMemsetAsync(view_pointers, 0, sizeof(int64_t), st));
        return MVF_OK;
    }
    MVF_REQUIRE(point_ids && image_ids && pixels && images && view_pointers && out_pixels &&
                (F == 0 || (features && out_features)), "mapping_from_dense: null buffer");
    SortWorkspace ws;
    if (!ws.carve(workspace, workspace_bytes, rows, num_points)) {
        set_error("mapping_from_dense: workspace too small (%zu < %zu)", workspace_bytes, ws.used);
        return MVF_E_WORKSPACE;
    }
    const dim3 block(256), grid_r(cdiv(rows, 256)), grid_r1(cdiv(rows + 1, 256));
    hipLaunchKernelGGL(fd_keys_kernel, grid_r, block, 0, st, (const i64 *)point_ids, (const i64 *)image_ids,
                       row2view, pixels, (i64)rows, (i64)num_images, ws.keys_a, ws.vals_a);
    const unsigned end_bit = 32u + (unsigned)sort_bits((u64)num_points * (u64)num_images - 1u);
    MVF_HIP(rocprim::radix_sort_pairs(ws.sort_tmp, ws.sort_tmp_bytes, (const u64 *)ws.keys_a, ws.keys_b,
                                      (const unsigned *)ws.vals_a, ws.vals_b, (size_t)rows, 0u, end_bit, st));
    hipLaunchKernelGGL(fd_flags_kernel, grid_r, block, 0, st, ws.keys_b, (i64)rows, dedupe, ws.flags);
    exclusive_scan<u64, u64, u64>(ws.flags, (i64)rows, ws.pos, ws.scan_blocks, (u64 *)nullptr, st);
    MVF_HIP(hipMemsetAsync(ws.point_views, 0, (size_t)(num_points + 1) * sizeof(unsigned), st));
    hipLaunchKernelGGL(fd_fill_kernel, grid_r1, block, 0, st, ws.keys_b, ws.vals_b, ws.flags, ws.pos, pixels,
                       (i64)rows, (i64)num_images, (i64 *)images, (i64 *)view_pointers, out_pixels,
                       ws.view_start, ws.point_views, (i64 *)counters);
    if (F > 0)
        hipLaunchKernelGGL(fd_features_kernel, dim3(cdiv(rows * F, 256)), block, 0, st, ws.vals_b, ws.flags,
                           ws.view_start, row2view, features, F, (const i64 *)counters, out_features);
    exclusive_scan<unsigned, i64, i64>(ws.point_views, (i64)num_points, (i64 *)pointers, ws.ptr_blocks,
                                       (i64 *)nullptr, st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_group_mean(
    const int64_t *keys, const float *feats, int32_t F, int64_t n, int64_t max_key, float *out,
    void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(n >= 0 && n < (1ll << 32) - 1 && F >= 1 && max_key >= 0, "group_mean: bad sizes");
    if (n == 0) return MVF_OK;
    MVF_REQUIRE(keys && feats && out, "group_mean: null buffer");
    SortWorkspace ws;
    if (!ws.carve(workspace, workspace_bytes, n, 1)) {
        set_error("group_mean: workspace too small (%zu < %zu)", workspace_bytes, ws.used);
        return MVF_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gm_keys_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const i64 *)keys, (i64)n,
                       ws.keys_a, ws.vals_a);
    MVF_HIP(rocprim::radix_sort_pairs(ws.sort_tmp, ws.sort_tmp_bytes, (const u64 *)ws.keys_a, ws.keys_b,
                                      (const unsigned *)ws.vals_a, ws.vals_b, (size_t)n, 0u,
                                      (unsigned)sort_bits((u64)max_key), st));
    hipLaunchKernelGGL(gm_mean_kernel, dim3(cdiv(n * F, 256)), dim3(256), 0, st, ws.keys_b, ws.vals_b, (i64)n,
                       feats, F, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_image_stats(
    const int64_t *view_pointers, const int64_t *images, const void *pixels, int32_t pixel_bytes,
    int64_t n_views, int64_t n_img, int64_t *stats, void *stream)
{
    MVF_REQUIRE(n_views >= 0 && n_img >= 0, "image_stats: bad sizes");
    MVF_REQUIRE(pixel_bytes == 2 || pixel_bytes == 8, "image_stats: pixels must be int16 or int64 (got %d bytes)",
                pixel_bytes);
    if (n_img == 0) return MVF_OK;
    MVF_REQUIRE(stats, "image_stats: null output");
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(256);
    hipLaunchKernelGGL(image_stats_init_kernel, dim3(cdiv(n_img * 5, 256)), block, 0, st, (i64 *)stats, (i64)n_img);
    if (n_views > 0) {
        MVF_REQUIRE(view_pointers && images && pixels, "image_stats: null buffer");
        const dim3 grid(cdiv(n_views, IS_VIEWS_PER_BLOCK));
        if (pixel_bytes == 2)
            hipLaunchKernelGGL(image_stats_kernel<int16_t>, grid, block, 0, st, (const i64 *)view_pointers,
                               (const i64 *)images, (const int16_t *)pixels, (i64)n_views, (i64)n_img,
                               (i64 *)stats);
        else
            hipLaunchKernelGGL(image_stats_kernel<i64>, grid, block, 0, st, (const i64 *)view_pointers,
                               (const i64 *)images, (const i64 *)pixels, (i64)n_views, (i64)n_img, (i64 *)stats);
    }
    hipLaunchKernelGGL(image_stats_finish_kernel, dim3(cdiv(n_img, 256)), block, 0, st, (i64 *)stats, (i64)n_img);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}
