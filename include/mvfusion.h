/*
 * mvfusion -- C ABI of the MI355X (gfx950) multi-view image->point fusion path.
 *
 * The reference implements this path in Python (numba on CPU, torch +
 * torch_scatter on GPU) and has no FFI of its own: every entry point below
 * replaces one Python-level function of
 * DeepViewAgg-Wildscenes/torch_points3d (cited per function) and is bound from
 * Python with ctypes (multimodal-fusion_amd/mvfusion/_lib.py).
 *
 * Conventions
 *  - Every pointer is a caller-owned DEVICE buffer unless its name starts with
 *    h_ (host).  Nothing is allocated or freed by the library; temporary
 *    storage comes from the caller's `workspace` (size from the matching
 *    *_workspace_bytes query, 256-byte aligned).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    work is enqueued asynchronously; data-dependent sizes are written to a
 *    device `counters` array that the caller reads after synchronising.
 *  - Return value: 0 = ok, negative = error (MVF_E_*); mvf_last_error() gives a
 *    thread-local message.  No exceptions cross the boundary.
 *  - Matrices are row-major.  Images are addressed [x][y] (x = column index in
 *    [0, W), y = row index), the layout of the reference's depth / index maps
 *    and masks.
 */
#ifndef MVFUSION_H
#define MVFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVF_VERSION 100

enum {
    MVF_OK = 0,
    MVF_E_INVALID = -1,    /* bad argument (shape, range, null pointer) */
    MVF_E_WORKSPACE = -2,  /* workspace too small */
    MVF_E_LAUNCH = -3      /* HIP launch / runtime error */
};

enum {
    MVF_CAM_KITTI360_PERSPECTIVE = 0,
    MVF_CAM_SCANNET = 1,
    MVF_CAM_KITTI360_FISHEYE = 2,
    MVF_CAM_S3DIS_EQUIRECTANGULAR = 3
};
enum { MVF_REDUCE_SUM = 0, MVF_REDUCE_MEAN = 1, MVF_REDUCE_MIN = 2, MVF_REDUCE_MAX = 3 };

/* One view.  Pinhole cameras: cam_to_world is the 4x4 extrinsic for
 * 'kitti360_perspective' and inverse(extrinsic) for 'scannet'
 * (core/multimodal/visibility.py:232-242).
 * 'kitti360_fisheye' (visibility.py:288-339): cam_to_world rows 0-2 hold the
 * extrinsic, its unused last row holds xi, k1, k2 in [12], [13], [14];
 * fx, fy, cx, cy hold gamma1, gamma2, u0, v0.
 * 's3dis_equirectangular' (visibility.py:150-182): cam_to_world[0..8] is the
 * row-major 3x3 rotation of pose_to_rotation_matrix_cpu(img_opk) (:57-90),
 * computed on the host; fx..cy are unused. */
typedef struct {
    float img_xyz[3];        /* camera centre used for the range test */
    float fx, fy, cx, cy;    /* intrinsic [0][0], [1][1], [0][2], [1][2] */
    float cam_to_world[16];
    double centre[3];        /* float64 camera position: radius sampling of MapImages */
    int32_t crop_off_x, crop_off_y; /* MapImages: crop offsets of this view */
} mvf_view;                  /* 128 bytes */

/* Parameters shared by all views of a call (VisibilityModel attributes,
 * visibility.py:1697-1705,1784-1790,1799-1801). */
typedef struct {
    int32_t W, H;            /* projection image size */
    int32_t crop_top, crop_bottom;
    int32_t camera;          /* MVF_CAM_* */
    int32_t exact;           /* SplattingVisibility.exact */
    double r_max, r_min;
    double voxel, k_swell, ln_d_swell; /* ln_d_swell = log(d_swell) on the host */
    float depth_threshold;   /* DepthBasedVisibility */
    int32_t pad_;
} mvf_vis_params;

int mvf_version(void);
const char *mvf_last_error(void);

/* Per-stage timing of the multi-kernel entry points with HIP events recorded on
 * the call's own stream (no counterpart in the reference; it exists for
 * bench.py's roofline figures).  State is per host thread.  After enabling,
 * each profiled call (mvf_map_images_exact) records one event per stage; once
 * the stream is synchronised mvf_profile_read fills h_ms[i] with the elapsed
 * milliseconds of stage i and h_names[i] (may be NULL) with its name, and
 * returns the number of stages (0 if nothing was recorded). */
#define MVF_PROFILE_MAX_STAGES 16
int mvf_profile_enable(int enable);
int mvf_profile_read(float *h_ms, const char **h_names, int capacity);

/* ---------------------------------------------------------------------- *
 * camera_projection  (visibility.py:478-538,592-623; all four cameras)
 * Keeps points with r_min < dist < r_max that project inside the (cropped)
 * image, in front of the camera, on a true mask pixel; outputs are compacted
 * in ascending point index.  Outputs need room for n entries.
 * counters[0] <- number kept.  mask: [W][H] bytes or NULL.
 * ---------------------------------------------------------------------- */
size_t mvf_camera_projection_workspace_bytes(int64_t n);
int mvf_camera_projection(
    const float *xyz, int64_t n, const mvf_view *h_view,
    const mvf_vis_params *h_params, const uint8_t *mask,
    int64_t *indices, float *dist, double *x_proj, double *y_proj,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* mvf_camera_projection for callers that cannot allocate device memory and
 * have no workspace to hand over (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same inputs, outputs and results as mvf_camera_projection; the
 * per-block counts of the compaction live in a 256 KB device array inside the
 * library (one copy per GPU, held until the process ends), so n is limited to
 * mvf_camera_projection_noalloc_max_points() = 2^24 points; a larger n is
 * rejected with MVF_E_INVALID.  The call allocates and frees nothing.  All
 * calls on one GPU share that array: calls that overlap on one GPU (different
 * streams or threads) race; serialise them.
 * The array's device address is looked up once per GPU (a symbol query that
 * may load the code object, and is no stream operation): that has to happen
 * OUTSIDE a graph capture.  mvf_camera_projection_noalloc_max_points() does
 * the look-up for the current device, so calling it once per GPU at set-up
 * makes every later mvf_camera_projection_noalloc call on that GPU capture-
 * safe (a first compute call outside capture does the same).  `stream` must
 * belong to the device that is current at the call. */
int64_t mvf_camera_projection_noalloc_max_points(void);
int mvf_camera_projection_noalloc(
    const float *xyz, int64_t n, const mvf_view *h_view,
    const mvf_vis_params *h_params, const uint8_t *mask,
    int64_t *indices, float *dist, double *x_proj, double *y_proj,
    int64_t *counters, void *stream);

/* ---------------------------------------------------------------------- *
 * visibility_from_splatting  (visibility.py:1073-1195,1288-1321; pinhole
 * splats :761-827).  Z-buffer of the m splats; ties in depth go to the lower
 * index; exact != 0 keeps one entry per seen point at its centre pixel (the
 * higher index wins a shared pixel).  Output order: ascending x then y.
 * out_* need room for `out_capacity` entries (m is enough when exact, W*Hc
 * always is).  counters[0] <- number of mappings, counters[1] <- number of
 * (splat, tile) pairs the call needed, counters[2] <- 1 if pair_capacity or
 * out_capacity was exceeded (outputs are then incomplete: call again with
 * larger buffers).
 * ---------------------------------------------------------------------- */
size_t mvf_splat_visibility_workspace_bytes(
    int64_t m, int32_t W, int32_t H, int64_t pair_capacity);
int mvf_splat_visibility(
    const double *x_proj, const double *y_proj, const float *dist, int64_t m,
    float fx, float fy, const mvf_vis_params *h_params, int64_t pair_capacity,
    int64_t *out_idx, int64_t *out_x, int64_t *out_y, int64_t out_capacity,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* The same for any camera of h_params->camera (visibility.py:1115-1132): the
 * splat boxes follow pinhole_splat_cpu (:761-827, fx / fy of h_view),
 * equirectangular_splat_cpu (:630-704, image size only) or fisheye_splat_cpu
 * (:876-953), which projects xyz[i] + (0, 0, swell * voxel / 2) again and
 * therefore needs the world coordinates xyz [m][3] of the splats; xyz may be
 * NULL for the other cameras.  Workspace as for mvf_splat_visibility. */
int mvf_splat_visibility_view(
    const double *x_proj, const double *y_proj, const float *dist, const float *xyz,
    int64_t m, const mvf_view *h_view, const mvf_vis_params *h_params,
    int64_t pair_capacity, int64_t *out_idx, int64_t *out_x, int64_t *out_y,
    int64_t out_capacity, int64_t *counters, void *workspace, size_t workspace_bytes,
    void *stream);

/* ---------------------------------------------------------------------- *
 * visibility_from_depth_map  (visibility.py:1361-1388) on a decoded [W][H]
 * float32 map (what read_s3dis_depth_map :1328-1358 returns).  Keeps
 * |map[int x][int y] - dist| <= depth_threshold, ascending index.
 * counters[0] <- number kept.
 * ---------------------------------------------------------------------- */
size_t mvf_depth_map_visibility_workspace_bytes(int64_t m);
int mvf_depth_map_visibility(
    const double *x_proj, const double *y_proj, const float *dist, int64_t m,
    const float *depth_map, int32_t W, int32_t H, float depth_threshold,
    int64_t *out_idx, double *out_x, double *out_y, int64_t *counters,
    void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- *
 * postprocess_features  (visibility.py:1548-1600): per mapped point
 * normalised depth, [linearity, planarity, scattering], [|cos| orientation],
 * pixel height.  idx: point index of each mapping (gathers xyz / attributes);
 * attribute pointers may be NULL.  out: [k][F] row-major,
 * F = 2 + #non-NULL attributes.
 * ---------------------------------------------------------------------- */
int mvf_mapping_features(
    const float *xyz, const int64_t *idx, const double *y_proj,
    const float *dist, int64_t k, const float *h_img_xyz,
    const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t H, double r_max, double r_min,
    float *out, void *stream);

/* ---------------------------------------------------------------------- *
 * MapImages._process, exact splat visibility, all V views of one sample in
 * one launch set (core/data_transform/multimodal/image.py:192-428 with
 * SplattingVisibility(exact=True), then ImageMapping.from_dense,
 * core/multimodal/image.py:1738-1805).
 *
 * xyz: [N][3] points ALREADY PERMUTED into the sampler's traversal order
 * (position in this array = local order used for depth ties and centre-pixel
 * conflicts, transforms.py:332,386); point_ids[i] = mapping id of the point at
 * position i; they must be distinct values in [0, num_points).
 * inv_ids: [num_points] position of each id (-1 if absent).
 * views: [V] device array of mvf_view.  cylinder != 0 tests the sampling
 * radius (r_max) in the xy plane only.  Attribute arrays follow the permuted
 * order; NULL = absent.  mask: [W][H] bytes or NULL.
 * Pixel conversion: pix // proj_upscale - crop offset, kept inside crop_size,
 * // downscale  (image.py:307-319); integer factors only.
 *
 * Outputs: pointers [num_points+1]; images / pixels(int16 [Vw][2]) /
 * features ([Vw][F]) with room for out_capacity views; view_seen [V] flags.
 * Since every (point, view) maps to one pixel the nested pixel pointers are
 * arange(Vw+1) and are not written.
 * counters (8 words) [0] <- Vw, [1] <- (splat, tile) pairs needed, [2] <- overflow flags
 * (bit 0: pair_capacity or out_capacity exceeded, bit 1: centre table full;
 * outputs are then incomplete: call again with larger sizes),
 * [3] <- number of views with at least one mapping.
 * centre_slots: size of the hash table (inside the workspace) that resolves
 * seen points sharing a centre pixel; rounded up to a power of two; about 2.5
 * times the expected number of mappings keeps probing short.
 * ---------------------------------------------------------------------- */
size_t mvf_map_images_workspace_bytes(
    int64_t n, int32_t V, int32_t W, int32_t H, int64_t num_points,
    int64_t pair_capacity, int64_t centre_slots);
int mvf_map_images_exact(
    const float *xyz, const int64_t *point_ids, const int64_t *inv_ids,
    int64_t n, int64_t num_points, const mvf_view *views, int32_t V,
    const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t centre_slots,
    int64_t pair_capacity, int64_t *pointers, int64_t *images, int16_t *pixels,
    float *features, int64_t out_capacity, int32_t *view_seen,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- *
 * mvf_map_images_exact for a batch of samples in one launch set: MapImages
 * applied to every sample (core/data_transform/multimodal/image.py:162-428)
 * followed by the collation of the per-sample mappings into one
 * ImageMappingBatch (CSRBatch.from_csr_list, core/multimodal/csr.py:347-416;
 * SameSettingImageBatch.from_data_list, core/multimodal/image.py:1329-1405).
 * The clouds of the samples are concatenated (xyz [n][3], every sample in its
 * own local order), and so are their views (views [V]); all samples share the
 * visibility parameters, image size and crop.  view_window [n][2] int32 =
 * {first view, number of views} of the sample of point position i;
 * view_first [V] = first view of the sample of view v; views_per_point = the
 * largest number of views of a sample.  inv_ids [num_points]: position of
 * every output point id (ids of sample s follow those of sample s - 1).
 * Outputs as mvf_map_images_exact over the concatenated ids: `images` holds
 * the rank of the view among the seen views of the whole batch, which is the
 * image index of the collated batch once every sample has dropped its unseen
 * views.  V <= 480, V * W * H < 2^32 - 1.
 * ---------------------------------------------------------------------- */
size_t mvf_map_images_batch_workspace_bytes(
    int64_t n, int32_t V, int32_t views_per_point, int32_t W, int32_t H,
    int64_t num_points, int64_t pair_capacity, int64_t centre_slots);
int mvf_map_images_exact_batch(
    const float *xyz, const int64_t *inv_ids, const int32_t *view_window,
    int64_t n, int64_t num_points, const mvf_view *views,
    const int32_t *view_first, int32_t V, int32_t views_per_point,
    const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t centre_slots,
    int64_t pair_capacity, int64_t *pointers, int64_t *images, int16_t *pixels,
    float *features, int64_t out_capacity, int32_t *view_seen,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- *
 * MapImages._process with SplattingVisibility(exact=False), all V views of one
 * sample in one launch set (core/data_transform/multimodal/image.py:192-428;
 * visibility_from_splatting_cpu without the exact step,
 * core/multimodal/visibility.py:1073-1162,1190-1195; ImageMapping.from_dense,
 * core/multimodal/image.py:1738-1805): every pixel a point wins in the z-buffer
 * is a mapping.  Arguments as for mvf_map_images_exact; proj_upscale and
 * downscale must be 1 (other factors merge pixels and go view by view).
 * Outputs: pointers [num_points+1]; images [Vw] and view_pointers [Vw+1] with
 * room for out_capacity views; pixels (int16 [P][2], ascending (x, y) inside a
 * view: the reference's order inside a view is unspecified) with room for
 * pix_capacity; features [Vw][F] = mean over the view's pixels.
 * mask_capacity: room of the pass's verdict buffer, in entries of 8 pixels:
 * every (point, view) pair that wins a pixel takes one entry per column of
 * its splat box and group of 8 image rows (8g .. 8g+7) the box reaches,
 * rounded up to a multiple of 8 (0 = half an entry per pixel of the views;
 * the workspace size follows it, so pass the same value to both functions).
 * counters (8 words): [0] <- Vw, [1] <- (splat, tile) pairs needed,
 * [2] <- overflow flags (bit 0: pair_capacity or out_capacity, bit 2:
 * pix_capacity, bit 3: mask_capacity), [3] <- views with at least one
 * mapping, [4] <- P, [5] <- verdict entries needed, [6] <- work items of the
 * pixel pass (informational).  After an overflow the outputs are incomplete:
 * call again with the capacities the counters name.
 * ---------------------------------------------------------------------- */
size_t mvf_map_images_splat_workspace_bytes(
    int64_t n, int32_t V, int32_t W, int32_t H, int64_t num_points, int64_t pair_capacity,
    int64_t mask_capacity);
int mvf_map_images_splat(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points,
    const mvf_view *views, int32_t V, const mvf_vis_params *h_params,
    int32_t cylinder, const uint8_t *mask, const float *linearity,
    const float *planarity, const float *scattering, const float *normals,
    int32_t crop_w, int32_t crop_h, int64_t pair_capacity, int64_t mask_capacity,
    int64_t *pointers, int64_t *images, int64_t *view_pointers, int16_t *pixels, float *features,
    int64_t out_capacity, int64_t pix_capacity, int32_t *view_seen,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* Profiling aid (no counterpart in the reference; bench.py's LDS-atomic
 * roofline): counts what the z-buffer kernel of the mvf_map_images_exact call
 * that last used `workspace` issued, from the tile bins it left there.  Pass
 * the same sizes as to that call.  d_out (device, caller zeroes it):
 * [0] += LDS atomic-minimum wave-instructions, [1] += lane-level atomics
 * (pixels + cells written), [2] += (splat, tile) entries, [3] += in-view
 * (point, view) pairs (the records the binning pass staged); d_out holds 4
 * 64-bit words. */
int mvf_profile_zbuffer_ops(
    void *workspace, size_t workspace_bytes, int64_t n, int32_t V,
    const mvf_vis_params *h_params, int64_t num_points, int64_t pair_capacity,
    int64_t centre_slots, uint64_t *d_out, void *stream);

/* ---------------------------------------------------------------------- *
 * MapImages._process with DepthBasedVisibility, all V views of one sample in
 * one launch set (core/data_transform/multimodal/image.py:192-428;
 * visibility_from_depth_map, core/multimodal/visibility.py:1361-1388; then
 * ImageMapping.from_dense, core/multimodal/image.py:1738-1805).
 * Arguments as for mvf_map_images_exact (the order of xyz does not matter here:
 * the depth test has no ties); depth_maps: [V][W][H] float32, decoded as
 * read_s3dis_depth_map returns them (visibility.py:1328-1358);
 * h_params->depth_threshold is the tolerance.  Both fused entry points take at
 * most 480 views per call.  counters (8 words): [0] <- Vw, [2] <- 1 if
 * out_capacity was exceeded, [3] <- number of views with at least one mapping.
 * ---------------------------------------------------------------------- */
size_t mvf_map_images_depth_workspace_bytes(int64_t num_points, int32_t V);
int mvf_map_images_depth(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points,
    const mvf_view *views, int32_t V, const mvf_vis_params *h_params,
    int32_t cylinder, const uint8_t *mask, const float *depth_maps,
    const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t *pointers, int64_t *images,
    int16_t *pixels, float *features, int64_t out_capacity, int32_t *view_seen,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- *
 * ImageMapping.from_dense (core/multimodal/image.py:1738-1805; lexargsort /
 * lexargunique, utils/multimodal.py:289-323; CSRData(dense=True) and
 * insert_empty_groups, core/multimodal/csr.py:54-79,158-172,197-229) in one
 * launch set: one 64-bit key per row, one stable radix sort, no host sort.
 * Rows are (point, image, pixel) triples; point_ids / image_ids / features are
 * given per row, or per source view when row2view [rows] (may be NULL) maps
 * every row to its view -- the dense expansion of
 * ImageMapping.select_points(mode='merge') (image.py:2245-2263) without the
 * repeat_interleave temporaries.  dedupe != 0 drops rows that repeat an
 * earlier (point, image, x, y) (lexargunique, image.py:2265-2268).
 * point_ids < num_points, image_ids < num_images, num_points * num_images
 * < 2^32, pixels non-negative.
 * Outputs: pointers [num_points + 1]; images, view_pointers (+1), out_pixels,
 * out_features ([views][F], mean over the view's kept rows in sorted order)
 * with room for `rows` entries.  counters[0] <- views, [1] <- pixels kept.
 * Pixels of a view come out in ascending (x, y), views of a point in ascending
 * image (the reference's order inside a view is unspecified).
 * ---------------------------------------------------------------------- */
size_t mvf_mapping_from_dense_workspace_bytes(int64_t rows, int64_t num_points);
int mvf_mapping_from_dense(
    const int64_t *point_ids, const int64_t *image_ids, const int32_t *row2view,
    const int16_t *pixels, const float *features, int32_t F, int64_t rows,
    int64_t num_points, int64_t num_images, int32_t dedupe, int64_t *pointers,
    int64_t *images, int64_t *view_pointers, int16_t *out_pixels,
    float *out_features, int64_t *counters, void *workspace, size_t workspace_bytes,
    void *stream);

/* out[r] = mean of feats over the rows that share keys[r] (0 <= keys <=
 * max_key), rows added in input order: the view features of
 * select_points(mode='merge') averaged over the views that land on the same
 * (point, image) pair and handed back to each of them (image.py:2229-2243).
 * Workspace: mvf_mapping_from_dense_workspace_bytes(n, 1). */
int mvf_group_mean(
    const int64_t *keys, const float *feats, int32_t F, int64_t n, int64_t max_key,
    float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------- *
 * Per-image statistics of a mapping in one pass over its views: stats
 * [n_img][5] = {mapped pixels, x min, x max, y min, y max} of every image
 * (all zero for an image without pixels).  Replaces the
 * repeat_interleave + torch_scatter.scatter_min / scatter_max of
 * ImageMapping.bounding_boxes (core/multimodal/image.py:1869-1879) and the
 * scatter_add / bounding-box areas of PickImagesFromMappingArea
 * (core/data_transform/multimodal/image.py:733-752); CropImageGroups
 * (:1040-1141) sizes its crops from the same boxes.
 * view_pointers [n_views + 1], images [n_views] (ids outside [0, n_img) are
 * skipped), pixels [P][2] int16 (pixel_bytes = 2) or int64 (8).
 * ---------------------------------------------------------------------- */
int mvf_image_stats(
    const int64_t *view_pointers, const int64_t *images, const void *pixels,
    int32_t pixel_bytes, int64_t n_views, int64_t n_img, int64_t *stats, void *stream);

/* ---------------------------------------------------------------------- *
 * torch_scatter.segment_csr(src, ptr, reduce) on dim 0 of a [rows][C] float32
 * matrix (modules/multimodal/pooling.py:63 and :289,295,519,525,628,799,819,
 * 863; core/multimodal/image.py:1777).  out: [nseg][C]; empty segment -> 0;
 * rows of a segment are folded in order (sum / mean are run-to-run
 * reproducible).  ptr: int64 [nseg+1], non-decreasing, ptr[nseg] <= rows.
 * ---------------------------------------------------------------------- */
int mvf_segment_csr(
    const float *src, int64_t rows, int32_t C, const int64_t *ptr, int64_t nseg,
    int32_t reduce, float *out, void *stream);

/* Gradient of segment_csr with respect to src, as torch_scatter's autograd
 * function computes it (torch_scatter is the reference's dependency for
 * pooling.py:63): sum -> the segment's gradient on every row, mean -> divided
 * by the row count, max / min -> on the first row holding the reduced value
 * (src and out = the forward's input and output, may be NULL for sum / mean).
 * Rows outside every segment are left untouched: pass a zeroed grad_src when
 * ptr[0] > 0 or ptr[nseg] < rows. */
int mvf_segment_csr_backward(
    const float *grad_out, const float *src, const float *out, int64_t rows, int32_t C,
    const int64_t *ptr, int64_t nseg, int32_t reduce, float *grad_src, void *stream);

/* Gradient of segment_softmax_csr (pooling.py:770-822) with respect to its
 * input, given the forward output y [rows][G]:
 * grad_src = y * (grad_out - sum over the segment of grad_out * y) / s, with
 * s = sqrt(rows of the segment) when scaling.  Terms of the order of the
 * reference's eps = 1e-12 are dropped.  Rows outside every segment are left
 * untouched. */
int mvf_segment_softmax_csr_backward(
    const float *grad_out, const float *y, int32_t G, const int64_t *ptr, int64_t nseg,
    int32_t scaling, float *grad_src, void *stream);

/* ----------------------------------------------------------------------
 * Differentiable tail of a dense layer: nn.BatchNorm1d + LeakyReLU on float32
 * rows z [rows][C] (contiguous), as the MLP blocks of
 * core/common_modules/base_modules.py:38-48 apply them behind their Linear
 * (FastBatchNorm1d :131-156 on 2D input is nn.BatchNorm1d).
 *
 * forward: training != 0 -> batch statistics (biased variance), and the
 *   running statistics (may be NULL) move by `momentum` towards the batch mean
 *   and the unbiased batch variance; training == 0 -> running statistics.
 *   y = act((z - mean) * invstd * weight + bias), act = LeakyReLU(slope) when
 *   act != 0, identity otherwise; weight / bias may be NULL (1 / 0).
 *   save_mean / save_invstd [C] receive the statistics used (for backward).
 * backward: grad_z, and grad_weight / grad_bias [C] (either may be NULL), from
 *   grad_y, the forward's input z and its saved statistics; the activation's
 *   sign is recomputed from z.  training must be the forward's.
 * Column sums are taken in float64 in a fixed order: results are the same on
 * every run.  col_sums (may be NULL): n_col_sums partial rows [2][C] of float64
 * whose sums over the rows are the column sums of z and z * z
 * (mvf_linear_rows_sums); the forward then does not read z for them.
 * workspace: mvf_batch_norm_act_workspace_bytes(C) bytes (the forward needs it
 * in train mode without col_sums only).
 * ---------------------------------------------------------------------- */
size_t mvf_batch_norm_act_workspace_bytes(int32_t C);
int mvf_batch_norm_act_forward(
    const float *z, int64_t rows, int32_t C, const float *weight, const float *bias,
    float *running_mean, float *running_var, int32_t training, float momentum, float eps,
    int32_t act, float slope, float *y, float *save_mean, float *save_invstd,
    const double *col_sums, int32_t n_col_sums, void *workspace, size_t workspace_bytes, void *stream);
int mvf_batch_norm_act_backward(
    const float *grad_y, const float *z, int64_t rows, int32_t C, const float *weight, const float *bias,
    const float *save_mean, const float *save_invstd, int32_t training, int32_t act, float slope,
    float *grad_z, float *grad_weight, float *grad_bias,
    void *workspace, size_t workspace_bytes, void *stream);

/* y = x W^T (+ bias) for contiguous float32 rows: x [rows][I], weight [O][I]
 * (torch.nn.Linear layout), bias [O] or NULL, y [rows][O]; I a multiple of 8,
 * O a multiple of 4, both at most 64 (mvf_linear_rows_supported), x and y
 * 16-byte aligned.  The output and the input gradient (weight = W^T) of the
 * Linear layers of the pools' MLPs and projections when a graph is recorded
 * (base_modules.py:38-48, pooling.py:406-452): a streaming pass on the float32
 * matrix cores with every row read and written once, in memory order. */
int mvf_linear_rows_supported(int32_t I, int32_t O);
int mvf_linear_rows(
    const float *x, int32_t I, const float *weight, const float *bias, int32_t O, int64_t rows,
    float *y, void *stream);
/* The same pass, which also leaves the column sums of y and y * y as
 * *n_col_sums partial rows [2][O] of float64 in col_sums (a buffer of
 * mvf_linear_rows_sums_bytes(O) bytes): what mvf_batch_norm_act_forward folds
 * the batch statistics from (its col_sums argument), so that a train-mode
 * Linear + BatchNorm1d reads the Linear's output once less. */
size_t mvf_linear_rows_sums_bytes(int32_t O);
int mvf_linear_rows_sums(
    const float *x, int32_t I, const float *weight, const float *bias, int32_t O, int64_t rows,
    float *y, double *col_sums, size_t col_sums_bytes, int32_t *n_col_sums, void *stream);

/* out[a][b] = sum over r of x[r][a] * y[r][b] (x: [rows][A], row stride ldx; y:
 * [rows][B], row stride ldy; out: [A][B], row stride ldo; float32): the weight
 * gradient of a torch.nn.Linear, grad_weight = grad_out^T . input, for the
 * Linear layers of the pools' MLPs and projections (base_modules.py:38-48,
 * pooling.py:406-452), whose reduction runs over every view of every point.
 * y == NULL with B == 1 stands for a column of ones: out[a][0] = sum over r of
 * x[r][a], the bias gradient.  float32 matrix cores, partial sums folded in a
 * fixed order (the same bits on every run).  workspace:
 * mvf_xty_workspace_bytes() bytes. */
size_t mvf_xty_workspace_bytes(void);
int mvf_xty(
    const float *x, int32_t ldx, int32_t A, const float *y, int32_t ldy, int32_t B, int64_t rows,
    float *out, int32_t ldo, void *workspace, size_t workspace_bytes, void *stream);

/* Attention-weighted segment sum with a graph (pooling.py:289-296, 519-526:
 * segment_csr(x_mod * expand_group_feat(A), csr_idx, 'sum')) and its gradients.
 * out[s][c] = sum over the rows r of segment s of x[r][c] * att[r][g(c)], with G
 * groups of C / G adjacent channels; rows of a segment are folded in order.
 * backward: grad_x[r][c] = grad_out[s(r)][c] * att[r][g(c)] and grad_att[r][g] =
 * sum over the channels of g of grad_out[s(r)][c] * x[r][c] (either may be
 * NULL); row2seg: int32 [rows] (mvf_csr_row_segments), a value outside
 * [0, nseg) marks a row of no segment: its gradients are zero.  Shapes: C / 4 and C / (4 G) powers of two, C <= 256; x, out and the
 * gradients 16-byte aligned and contiguous. */
int mvf_weighted_segment_sum(
    const float *x, const float *att, int32_t C, int32_t G, const int64_t *ptr, int64_t nseg,
    float *out, void *stream);
int mvf_weighted_segment_sum_backward(
    const float *grad_out, const float *x, const float *att, int32_t C, int32_t G,
    const int32_t *row2seg, int64_t rows, int64_t nseg, float *grad_x, float *grad_att, void *stream);

/* gather_csr (pooling.py:825-853): out[r] = src[segment of r], [rows][C]. */
int mvf_gather_csr(
    const float *src, int32_t C, const int64_t *ptr, int64_t nseg, int64_t rows,
    float *out, void *stream);

/* segment_softmax_csr (pooling.py:770-822): softmax inside CSR segments of a
 * [rows][G] matrix, optional division of the centred scores by sqrt(segment
 * size), eps added to the denominator. */
int mvf_segment_softmax_csr(
    const float *src, int32_t G, const int64_t *ptr, int64_t nseg, float eps,
    int32_t scaling, float *out, void *stream);

/* Both pooling levels of the fusion in one pass over src:
 *   out_x = segment_csr(segment_csr(src, inner_ptr, inner_reduce), outer_ptr, reduce_x)
 * i.e. BimodalCSRPool(reduce_x)(.., BimodalCSRPool(inner_reduce)(.., src, .., atomic_csr_indexing),
 * .., view_csr_indexing) for x = a and, when out_b is not null, x = b
 * (modules/multimodal/pooling.py:14-71 applied at the atomic and at the view level).
 * inner_ptr: [n_inner + 1] rows of src per inner segment; outer_ptr: [n_outer + 1]
 * inner segments per outer segment.  Results equal the two separate calls bit for bit. */
int mvf_segment_csr_two_level(
    const float *src, int64_t rows, int32_t C, const int64_t *inner_ptr, int64_t n_inner,
    int32_t inner_reduce, const int64_t *outer_ptr, int64_t n_outer, int32_t reduce_a, float *out_a,
    int32_t reduce_b, float *out_b, void *stream);

/* Feature-map gather fused with the atomic-level pool:
 *   out[r][c] = reduce over p in [view_ptr[r], view_ptr[r+1]) of x[images[r]][c][pixels[p][1]][pixels[p][0]]
 * for x [B][C][H][W] float32, i.e. BimodalCSRPool(mode)(x[mapping.feature_map_indexing],
 * mapping.atomic_csr_indexing) without the [P][C] intermediate
 * (core/multimodal/image.py:1272-1297, 1881-1902; modules/multimodal/pooling.py:63).
 * pixels are (x = width index, y = height index) int16 pairs; the caller checks
 * that images / pixels index inside x. */
int mvf_gather_pool(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *images,
    const int64_t *view_ptr, const int16_t *pixels, int64_t rows, int32_t reduce, float *out,
    void *stream);

/* X[mappings.feature_map_indexing] (core/multimodal/image.py:1295,1881-1895):
 * out[p][c] = x[batch[p]][c][pixels[p][1]][pixels[p][0]] for x of shape
 * [B][C][H][W]; batch [rows], pixels [rows][2] = (column, row).  The caller
 * guarantees batch < B and pixels inside the map. */
int mvf_gather_features(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *batch,
    const int16_t *pixels, int64_t rows, float *out, void *stream);

/* Gradient of X[mappings.feature_map_indexing] with respect to X (what autograd
 * attaches to the indexing of core/multimodal/image.py:1295): grad_x [B][C][H][W]
 * (the caller zeroes it) += grad_out [rows][C] at the rows' pixels.  Rows that
 * share a pixel are added atomically (order dependent in the last bits). */
int mvf_gather_features_backward(
    const float *grad_out, int32_t B, int32_t C, int32_t H, int32_t W, const int64_t *batch,
    const int16_t *pixels, int64_t rows, float *grad_x, void *stream);

/* Gradient of sparse_interpolation (core/multimodal/image.py:105-170) with
 * respect to the feature maps: grad_x (zeroed by the caller) += the four
 * bilinear weights of every query times its gradient row. */
int mvf_sparse_interpolation_backward(
    const float *grad_out, int32_t B, int32_t C, int32_t H, int32_t W, const float *coords,
    const int64_t *batch, int64_t rows, int32_t padding_mode, float *grad_x, void *stream);

/* sparse_interpolation (core/multimodal/image.py:105-170): bilinear sampling
 * of x [B][C][H][W] at coords [rows][2] = (row, column) in [0, 1] on the map
 * padded by one pixel.  padding_mode: 0 'border', 1 'zeros', 2 'reflection'.
 * out: [rows][C]. */
int mvf_sparse_interpolation(
    const float *x, int32_t B, int32_t C, int32_t H, int32_t W, const float *coords,
    const int64_t *batch, int64_t rows, int32_t padding_mode, float *out, void *stream);

/* ------------------------------------------------------------------------- */
/* attentive pools: QKVBimodalCSRPool / GroupBimodalCSRPool / DeepSetFeat       */
/* ------------------------------------------------------------------------- */

/* One layer of MLP(channels, bias=False) in eval mode
 * (core/common_modules/base_modules.py:38-48: Linear, FastBatchNorm1d
 * :131-156, LeakyReLU(0.2)) or a plain nn.Linear (all bn_* NULL, act = 0).
 * All pointers are device pointers. */
typedef struct mvf_mlp_layer {
    const float *weight;    /* [cout][cin], torch Linear.weight */
    const float *bias;      /* [cout] or NULL */
    const float *bn_weight; /* BatchNorm1d weight / bias / running_mean / running_var, */
    const float *bn_bias;   /* [cout] each; all four NULL = no batch norm */
    const float *bn_mean;
    const float *bn_var;
    float bn_eps;
    float slope;            /* LeakyReLU negative slope, used when act != 0 */
    int32_t cin, cout, act, pad_;
} mvf_mlp_layer;

/* A chain of 1..3 layers on the fp32 matrix cores, y = L_n(..L_1(in)), with
 * in[row] = [xa[ga(row)] | xb[gb(row)]] (mode 0; xb may be NULL),
 * xa + xb (mode 1) or [xa | xa + xb] (mode 2): the fusions of
 * DeepSetFeat.forward (pooling.py:633-638, 671-680); ga / gb are optional row
 * gathers (gather_csr of a per-segment tensor, pooling.py:825-853, or
 * torch.repeat_interleave, pooling.py:489-501).  Every cout <= 64 and every
 * cin after the first <= 64; the first cin is free.  `layers` is a host array. */
int mvf_mlp_forward(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    float *out, int32_t ldo, void *stream);

/* Arithmetic of the layer chains.  mode 1 (default): chains of two layers that are both
 * wider than 32 outputs, on at most 64 aligned and ungathered input columns (E_mod,
 * pooling.py:757, 482-484, and the value encoders of the grouped pools, :272-300), and
 * chains of three layers of at most 32 outputs on at most 64 aligned input columns (the
 * key / query branches, pooling.py:486-501), run on
 * the bf16 matrix cores with every float32 operand split exactly into three bf16 parts and
 * the six significant part products accumulated in float32: the accuracy of the float32
 * chain (the reference computes these layers with a float32 BLAS, whose summation order is
 * not part of its contract) at 3/8 of its matrix-pipe time.  (Inputs and weights of a
 * magnitude of 2^127 and above round to infinity in their first bf16 part and give NaN
 * where the float32 chain gives a large finite value or infinity.)  mode 0: every chain on the
 * float32 matrix cores (exact fmaf chains).  The environment variable
 * MVF_MLP_PRECISION=fp32 selects mode 0 at load time.  Process-wide; no counterpart in the
 * reference. */
int mvf_set_mlp_precision(int32_t mode);
int mvf_get_mlp_precision(void);

/* The same chain with a dot-product epilogue: the last layer's output y
 * [rows][cout <= 32] is not stored; instead
 *   dot_out[r] = sum_c y[r][c] * queries[row2seg[r]][c]     (/ divisor when divisor != 0)
 * i.e. keys = K(...) and compatibilities = (keys * queries).sum / sqrt(D) of
 * QKVBimodalCSRPool.forward with one group (pooling.py:486-508) without the
 * keys ever leaving the chip.  Products and sums are rounded one by one, the
 * order of the sum differs from the reference's (within fp32 tolerance). */
int mvf_mlp_forward_dot(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int32_t mode, int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    const float *queries, int32_t ldq, const int32_t *row2seg, float divisor, float *dot_out,
    void *stream);

/* The same chain (two layers of equal output width <= 64: E_mod, pooling.py:757,
 * 482-484; mode 0 inputs without row gathers, 16-byte aligned) with the attention-weighted segment
 * sum as epilogue: the encoded view features are not stored; instead
 *   pool_out[s][c] = sum over the rows r of segment s of y[r][c] * att[r][group(c)]
 * i.e. pooling.py:519-530 / :295-300 without the [rows][cout] values leaving
 * the chip.  row2seg maps rows to segments (rows grouped by segment,
 * ascending).  att is [rows][G], G <= 8 groups of cout / G consecutive channels
 * (cout % G == 0, a multiple of 4 channels per group unless G == 1), and
 * already holds att * gate.  att and pool_out must come from
 * mvf_attention_pool(x_mod = NULL), which folds the gate that way and zeroes
 * the rows of empty segments.  Segments are folded in row order inside tiles
 * of 32 rows (the wide chains: as a balanced tree over the tile's rows); a
 * segment that crosses a tile boundary leaves one partial sum per tile in
 * `workspace` (mvf_mlp_forward_pool_workspace_bytes(rows, cout)), and a second
 * launch adds those up in tile order: no atomics, the same bits on every run.
 * ptr [nseg + 1] and row2seg [rows] describe the same segments.  Results agree
 * with mvf_attention_pool's to float32 rounding (bit for bit when there is no
 * gate and the segment lies inside one tile of the float32 chain). */
size_t mvf_mlp_forward_pool_workspace_bytes(int64_t rows, int32_t cout);
int mvf_mlp_forward_pool(
    const float *xa, int32_t lda, int32_t cols_a, const int32_t *gather_a,
    const float *xb, int32_t ldb, int32_t cols_b, const int32_t *gather_b,
    int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers,
    const float *att, int32_t G, const int32_t *row2seg, const int64_t *ptr, int64_t nseg,
    float *pool_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same chain (two layers of equal output width <= 64: DeepSetFeat.mlp_elt_1,
 * pooling.py:633; mode 0 inputs without row gathers, 16-byte aligned rows in and
 * out) that stores its rows AND their per-segment maximum,
 *   seg_out[s][c] = max over the rows r of segment s of out[r][c]     (0 for an empty segment)
 * i.e. segment_csr(mlp_elt_1(x), csr_idx, reduce='max') of pooling.py:628-633
 * without reading the rows back.  ptr [nseg + 1] and row2seg [rows] describe the
 * same segments.  Bit-identical to mvf_mlp_forward followed by mvf_segment_csr
 * (segments crossing a 32-row tile are combined with atomic maxima, which do
 * not depend on order). */
int mvf_mlp_forward_segmax(
    const float *xa, int32_t lda, int32_t cols_a, const float *xb, int32_t ldb, int32_t cols_b,
    int64_t rows, const mvf_mlp_layer *layers, int32_t nlayers, float *out, int32_t ldo,
    const int64_t *ptr, int64_t nseg, const int32_t *row2seg, float *seg_out, void *stream);

/* The attention tail shared by QKVBimodalCSRPool.forward (pooling.py:503-530)
 * and GroupBimodalCSRPool.forward (pooling.py:284-300):
 *   comp[r][g] = sum_k keys[r][g D + k] * queries[seg(r)][g D + k]   (/ sqrt(D) if dim_scaling)
 *                (seg(r) = row2seg[r]; row2seg == NULL: queries has one row per key row, the
 *                 use_mod_q case, pooling.py:487-494; keys == NULL: comp already holds the
 *                 compatibilities)
 *   att        = segment_softmax_csr(comp, ptr, scaling = group_scaling)
 *   out[s][c]  = sum_r x_mod[r][c] * att[r][chan2group[c]]
 *   gating:      gate[s][g] = tanh(relu(max_r comp[r][g] * gate_w[g] + gate_b[g])),
 *                out[s][c] *= gate[s][chan2group[c]]               (Gating, pooling.py:702-723)
 * comp / att are [rows][G], gate is [nseg][G]; they are outputs the caller
 * may keep (the modules' save_last tensors).  chan2group may be NULL when G == 1.
 * x_mod == NULL with rows > 0: comp, att and gate are computed and `out` [nseg][C]
 * is prepared for mvf_mlp_forward_pool, which then produces the sum; att comes
 * back times the gate of its segment (needs G <= 8, C % G == 0, (C / G) % 4 == 0
 * and a 16-byte aligned `out`). */
int mvf_attention_pool(
    const float *keys, int32_t ldk, const float *queries, int32_t ldq, const int32_t *row2seg,
    const float *x_mod, int32_t C, const int64_t *ptr, int64_t nseg, int64_t rows, int32_t G, int32_t D,
    int32_t dim_scaling, int32_t group_scaling, float eps, const int32_t *chan2group,
    const float *gate_w, const float *gate_b, int32_t gating,
    float *comp, float *att, float *gate, float *out, void *stream);

/* row -> segment index of a CSR pointer array (the dense index that
 * pooling.py:306-308 builds with arange().repeat_interleave()). */
int mvf_csr_row_segments(const int64_t *ptr, int64_t nseg, int32_t *row2seg, void *stream);

/* HeuristicBimodalCSRPool.forward (pooling.py:129-152): out[s] = x_mod[row of
 * the segment's largest (pick_max) or smallest value of feat], zeros for an
 * empty segment; feat is one column of the mapping features, stride ldf.
 * picked (may be NULL): [nseg] the chosen row of every segment, -1 for an empty
 * one (the rows the gradient of x_mod goes to). */
int mvf_segment_pick(
    const float *x_mod, int32_t C, const float *feat, int32_t ldf, const int64_t *ptr,
    int64_t nseg, int32_t pick_max, float *out, int64_t *picked, void *stream);

/* ---------------------------------------------------------------------- *
 * NeighborhoodBasedMappingFeatures
 * (core/data_transform/multimodal/image.py:431-612).
 * mvf_knn: the k <= 64 nearest neighbours of every point among the n points
 * (the point itself included, first), neighbors [n][k] ascending in distance;
 * exact: float64 squared distances of the float32 coordinates, equal
 * distances by ascending id.  ids [n] (NULL = row index) names the rows: the
 * output holds ids, row r of neighbors belongs to row r of xyz.  Any row order
 * gives the same result; rows sorted along a space-filling curve make it
 * several times faster.  box_lo / box_hi [ceil(n / 256)][3] (or both NULL):
 * bounding boxes of the rows 256 c .. 256 c + 255; with them, chunks that are
 * too far from a block of queries to matter are skipped.  The reference searches with FAISS (IVF,
 * approximate) or KeOps argKmin (float32, exact), :497-518.
 * mvf_mapping_occlusion: for each view-level row r (point row_point[r], image
 * images[r]): (1 + #{j < k : seen[neighbors[point][j]][image]}) / (k + 1)
 * (:567-596); seen [n_points][n_images] bytes; out[r * out_stride].
 * ---------------------------------------------------------------------- */
int mvf_knn(const float *xyz, const int64_t *ids, int64_t n, int32_t k, const float *box_lo,
            const float *box_hi, int64_t *neighbors, void *stream);
int mvf_mapping_occlusion(
    const int64_t *neighbors, int32_t k_stride, int32_t k, const int64_t *row_point,
    const int64_t *images, int64_t rows, const uint8_t *seen, int64_t n_images, float *out,
    int32_t out_stride, void *stream);

/* ---------------------------------------------------------------------- *
 * Local point order of MapImages (host code, no device work).
 * The reference's radius samplers (core/data_transform/transforms.py:326,332,
 * 380,386) return the points of a sample in the order of an sklearn
 * KDTree(leaf_size=50).idx_array; depth ties and shared centre pixels are
 * resolved on that order (core/multimodal/visibility.py:1157,1183-1187).
 * mvf_kdtree_order rebuilds that index array: h_pos [n] HOST rows of `stride`
 * floats whose first dims (2: cylinder sampling, 3: sphere sampling) columns
 * are the coordinates (numbers, no NaN), h_order [n] HOST int64.  threads < 1:
 * as many as the host offers (at most 16).  Same permutation as sklearn for
 * any thread count.
 * ---------------------------------------------------------------------- */
int mvf_kdtree_order(const float *h_pos, int64_t n, int32_t dims, int64_t stride, int64_t leaf_size,
                     int64_t *h_order, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* MVFUSION_H */
