/*
 * CPU oracle for the multi-view image->point fusion path.  TEST INFRASTRUCTURE
 * ONLY: it is the checker for the HIP kernels (tests/, smoke) and the timed
 * "port" CPU baseline of bench.py.  Nothing in the shipped package links or
 * calls it.
 *
 * It restates, loop for loop and dtype for dtype, the reference's CPU
 * (numba + numpy) algorithm.  Citations are relative to
 * /root/reference/DeepViewAgg-Wildscenes/torch_points3d/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see
 * oracle/Makefile).  -ffp-contract=off matters: every float32 operation of the
 * reference is separately rounded except the 3x3 rotation, which numpy hands
 * to an sgemm whose result equals an explicit fmaf chain in k order 0,1,2.
 *
 * Parity pin: the .npz fixtures under tests/golden, produced by tests/golden/make_golden.py
 * running the reference's own source files in the build container.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

enum { CAM_KITTI360_PERSPECTIVE = 0, CAM_SCANNET = 1 };
enum { RED_SUM = 0, RED_MEAN = 1, RED_MIN = 2, RED_MAX = 3 };

/* ------------------------------------------------------------------------ *
 * camera_projection_cpu  (core/multimodal/visibility.py:478-538)
 *   norm_cpu :129-137, pinhole_projection_cpu :219-252,
 *   field_of_view_cpu :395-435.
 * cam_to_world: row-major 4x4 float32.  For 'kitti360_perspective' it is the
 * extrinsic itself (:239); for 'scannet' the caller passes
 * inv(extrinsic) computed in float32 (:233), so this file needs no LAPACK.
 * mask: [W][H] bytes or NULL.  r_min / r_max are compared in double against
 * the float32 distance (numba promotes the float64 scalar bound; identical to
 * a float32 compare whenever the bound is float32-representable).
 * Returns the number of points kept; outputs are filled in ascending index.
 * ------------------------------------------------------------------------ */
i64 orc_camera_projection(
    i64 n, const float *xyz, const float *img_xyz, const float *cam_to_world,
    float fx, float fy, float cx, float cy, const unsigned char *mask,
    int W, int H, int crop_top, int crop_bottom, double r_max, double r_min,
    int camera, i64 *indices, float *dist_out, double *x_out, double *y_out)
{
    const float *M = cam_to_world;
    const float T0 = M[3], T1 = M[7], T2 = M[11];
    i64 m = 0;
    for (i64 i = 0; i < n; ++i) {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        /* norm_cpu(xyz - img_xyz): sqrt(((dx^2 + dy^2) + dz^2)), float32 */
        const float dx = px - img_xyz[0], dy = py - img_xyz[1], dz = pz - img_xyz[2];
        const float sx = dx * dx, sy = dy * dy, sz = dz * dz;
        const float s01 = sx + sy;
        const float s = s01 + sz;
        const float d = sqrtf(s);
        if (!(r_min < (double)d && (double)d < r_max)) continue;

        float p0, p1, p2;
        if (camera == CAM_KITTI360_PERSPECTIVE) {
            /* p = R.T @ (xyz - T).T  (:240-242) */
            const float e0 = px - T0, e1 = py - T1, e2 = pz - T2;
            p0 = fmaf(M[8], e2, fmaf(M[4], e1, M[0] * e0));
            p1 = fmaf(M[9], e2, fmaf(M[5], e1, M[1] * e0));
            p2 = fmaf(M[10], e2, fmaf(M[6], e1, M[2] * e0));
        } else {
            /* p = R @ xyz.T + T  (:234-236) */
            const float q0 = fmaf(M[2], pz, fmaf(M[1], py, M[0] * px));
            const float q1 = fmaf(M[6], pz, fmaf(M[5], py, M[4] * px));
            const float q2 = fmaf(M[10], pz, fmaf(M[9], py, M[8] * px));
            p0 = q0 + T0; p1 = q1 + T1; p2 = q2 + T2;
        }
        /* x = p0 * fx / p2 + cx  (:246-247), each op rounded to float32 */
        const float ax = p0 * fx, bx = ax / p2, xf = bx + cx;
        const float ay = p1 * fy, by = ay / p2, yf = by + cy;
        const double x = (double)xf, y = (double)yf, z = (double)p2;

        /* field_of_view_cpu */
        if (!(0.0 <= x)) continue;
        if (!((double)crop_top <= y)) continue;
        if (!(x < (double)W)) continue;
        if (!(y < (double)(H - crop_bottom))) continue;
        if (!(0.0 < z)) continue;
        if (mask) {
            const uint32_t xi = (uint32_t)floor(x), yi = (uint32_t)floor(y);
            if (!mask[(size_t)xi * H + yi]) continue;
        }
        indices[m] = i; dist_out[m] = d; x_out[m] = x; y_out[m] = y;
        ++m;
    }
    return m;
}

/* float64 -> rint -> float32 -> int32, as np.round(v, 0, out_f32).astype(int32) */
static inline int32_t round_f64_to_i32(double v)
{
    const float f = (float)rint(v);
    /* x86 cvttss2si semantics for out-of-range / NaN: INT_MIN */
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

/* ------------------------------------------------------------------------ *
 * pinhole_splat_cpu  (visibility.py:761-827).  splat: [m][4] int32 =
 * (x_a, x_b, y_a, y_b), NOT yet shifted by crop_top.
 * ------------------------------------------------------------------------ */
void orc_pinhole_splat(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    float fx, float fy, int W, int H, int crop_top, int crop_bottom,
    double voxel, double k_swell, double d_swell, int32_t *splat)
{
    const double ln_d = log(d_swell);
    const int32_t x_min = 0, x_max = W, y_min = crop_top, y_max = H - crop_bottom;
    for (i64 i = 0; i < m; ++i) {
        const double d = (double)dist[i];
        const double swell = (1.0 + k_swell * exp(-d / ln_d)) * voxel / d;
        const double wx = swell * (double)fx, wy = swell * (double)fy;
        int32_t xa = round_f64_to_i32(x_proj[i] - wx / 2);
        int32_t xb = round_f64_to_i32(x_proj[i] + wx / 2 + 1);
        int32_t ya = round_f64_to_i32(y_proj[i] - wy / 2);
        int32_t yb = round_f64_to_i32(y_proj[i] + wy / 2 + 1);
        splat[4 * i + 0] = clampi(xa, x_min, x_max - 1);
        splat[4 * i + 1] = clampi(xb, x_min + 1, x_max);
        splat[4 * i + 2] = clampi(ya, y_min, y_max - 1);
        splat[4 * i + 3] = clampi(yb, y_min + 1, y_max);
    }
}

/* ------------------------------------------------------------------------ *
 * visibility_from_splatting_cpu  (visibility.py:1073-1195), pinhole cameras.
 * Sequential z-buffer over a [W][Hc] depth map, strict '<', points visited in
 * ascending index (:1148-1162); exact=1 re-stamps only the splat centres of
 * the seen points, ascending, later overwrites earlier (:1168-1187); output in
 * np.where order of the [W][Hc] map = sorted by x then y (:1190-1195), y
 * expressed in the non-cropped image.  Outputs need room for
 * min(m, W*Hc) (exact) or W*Hc (non exact) entries.  Returns the count, or -1
 * on allocation failure.
 * ------------------------------------------------------------------------ */
i64 orc_visibility_from_splatting(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    float fx, float fy, int W, int H, int crop_top, int crop_bottom,
    double voxel, double k_swell, double d_swell, int exact,
    i64 *out_idx, i64 *out_x, i64 *out_y)
{
    const int Hc = H - crop_bottom - crop_top;
    const size_t npix = (size_t)W * Hc;
    int32_t *splat = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(m > 0 ? m : 1));
    float *depth = (float *)malloc(sizeof(float) * npix);
    i64 *idmap = (i64 *)malloc(sizeof(i64) * npix);
    if (!splat || !depth || !idmap) { free(splat); free(depth); free(idmap); return -1; }

    orc_pinhole_splat(m, x_proj, y_proj, dist, fx, fy, W, H, crop_top,
                      crop_bottom, voxel, k_swell, d_swell, splat);

    float dmax = dist[0];
    for (i64 i = 1; i < m; ++i) if (dist[i] > dmax) dmax = dist[i];
    const float init = (dmax + 1) + 1;
    for (size_t p = 0; p < npix; ++p) { depth[p] = init; idmap[p] = -1; }

    for (i64 i = 0; i < m; ++i) {
        const float d = dist[i];
        const int xa = splat[4 * i], xb = splat[4 * i + 1];
        const int ya = splat[4 * i + 2] - crop_top, yb = splat[4 * i + 3] - crop_top;
        for (int x = xa; x < xb; ++x) {
            float *drow = depth + (size_t)x * Hc;
            i64 *irow = idmap + (size_t)x * Hc;
            for (int y = ya; y < yb; ++y) {
                if (d < drow[y]) { drow[y] = d; irow[y] = i; }
            }
        }
    }

    if (exact) {
        unsigned char *seen = (unsigned char *)calloc((size_t)(m > 0 ? m : 1), 1);
        if (!seen) { free(splat); free(depth); free(idmap); return -1; }
        for (size_t p = 0; p < npix; ++p) if (idmap[p] >= 0) seen[idmap[p]] = 1;
        for (size_t p = 0; p < npix; ++p) idmap[p] = -1;
        for (i64 i = 0; i < m; ++i) {
            if (!seen[i]) continue;
            const int x = (int32_t)x_proj[i];
            const int y = (int32_t)y_proj[i] - crop_top;
            idmap[(size_t)x * Hc + y] = i;
        }
        free(seen);
    }

    i64 k = 0;
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < Hc; ++y) {
            const i64 id = idmap[(size_t)x * Hc + y];
            if (id >= 0) { out_idx[k] = id; out_x[k] = x; out_y[k] = y + crop_top; ++k; }
        }
    free(splat); free(depth); free(idmap);
    return k;
}

/* ------------------------------------------------------------------------ *
 * visibility_from_depth_map  (visibility.py:1361-1388) on an already decoded
 * [W][H] float32 map (read_s3dis_depth_map :1328-1358 yields raw/512 with
 * empties = -1).  Keeps |map[int x, int y] - dist| <= threshold, float32.
 * ------------------------------------------------------------------------ */
i64 orc_visibility_from_depth_map(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    const float *depth_map, int W, int H, float threshold, i64 *out_idx)
{
    (void)W;
    i64 k = 0;
    for (i64 i = 0; i < m; ++i) {
        const i64 x = (i64)x_proj[i], y = (i64)y_proj[i];
        const float real = depth_map[(size_t)x * H + y];
        const float diff = real - dist[i];
        if (fabsf(diff) <= threshold) out_idx[k++] = i;
    }
    return k;
}

/* ------------------------------------------------------------------------ *
 * postprocess_features  (visibility.py:1548-1600) for k mapped points.
 * Column order: normalised depth, [linearity], [planarity], [scattering],
 * [orientation if normals], pixel height.  lin/pla/sca/normals are already
 * gathered per mapped point (or NULL).  xyz_to_img = xyz[idx] - img_xyz.
 * out: [k][F] float32 row-major.  Returns F.
 * ------------------------------------------------------------------------ */
int orc_postprocess_features(
    i64 k, const float *xyz_to_img, const double *y_proj, const float *dist,
    const float *lin, const float *pla, const float *sca, const float *normals,
    int H, double r_max, double r_min, float *out)
{
    const int F = 2 + (lin != 0) + (pla != 0) + (sca != 0) + (normals != 0);
    const float lo = (float)r_min, hi = (float)(r_max + 1e-4), eps = (float)1e-4;
    for (i64 i = 0; i < k; ++i) {
        float *o = out + (size_t)i * F;
        int c = 0;
        const float t = dist[i] - lo;
        o[c++] = t / hi;
        if (lin) o[c++] = lin[i];
        if (pla) o[c++] = pla[i];
        if (sca) o[c++] = sca[i];
        if (normals) {
            const float den = dist[i] + eps;
            const float u0 = xyz_to_img[3 * i] / den, u1 = xyz_to_img[3 * i + 1] / den,
                        u2 = xyz_to_img[3 * i + 2] / den;
            const float a = u0 * normals[3 * i], b = u1 * normals[3 * i + 1],
                        cc = u2 * normals[3 * i + 2];
            const float ab = a + b;
            o[c++] = fabsf(ab + cc);
        }
        o[c++] = (float)(y_proj[i] / (double)H);
    }
    return F;
}

/* ------------------------------------------------------------------------ *
 * torch_scatter.segment_csr(src, ptr, reduce) on dim 0 of a [rows][C] float32
 * matrix (call sites: modules/multimodal/pooling.py:63,289,295,519,525,628,
 * 799,819,863; core/multimodal/image.py:1777).  Published CPU semantics:
 * rows of a segment are folded in order; empty segment -> 0 for every
 * reduction; mean = sum / count.
 * ------------------------------------------------------------------------ */
void orc_segment_csr(
    const float *src, int C, const i64 *ptr, i64 nseg, int mode, float *out)
{
    for (i64 s = 0; s < nseg; ++s) {
        const i64 a = ptr[s], b = ptr[s + 1];
        float *o = out + (size_t)s * C;
        if (b <= a) { for (int c = 0; c < C; ++c) o[c] = 0.0f; continue; }
        for (int c = 0; c < C; ++c) o[c] = src[(size_t)a * C + c];
        for (i64 r = a + 1; r < b; ++r) {
            const float *x = src + (size_t)r * C;
            if (mode == RED_MAX)      { for (int c = 0; c < C; ++c) if (x[c] > o[c]) o[c] = x[c]; }
            else if (mode == RED_MIN) { for (int c = 0; c < C; ++c) if (x[c] < o[c]) o[c] = x[c]; }
            else                      { for (int c = 0; c < C; ++c) o[c] = o[c] + x[c]; }
        }
        if (mode == RED_MEAN) {
            const float cnt = (float)(b - a);
            for (int c = 0; c < C; ++c) o[c] = o[c] / cnt;
        }
    }
}

/* ------------------------------------------------------------------------ *
 * segment_softmax_csr  (pooling.py:770-822): centre on the segment max,
 * optionally divide by sqrt(segment size), exp, divide by (segment sum + eps).
 * src/out: [rows][G] float32.
 * ------------------------------------------------------------------------ */
void orc_segment_softmax_csr(
    const float *src, int G, const i64 *ptr, i64 nseg, float eps, int scaling,
    float *out)
{
    for (i64 s = 0; s < nseg; ++s) {
        const i64 a = ptr[s], b = ptr[s + 1];
        if (b <= a) continue;
        const float sq = sqrtf((float)(b - a));
        for (int g = 0; g < G; ++g) {
            float mx = src[(size_t)a * G + g];
            for (i64 r = a + 1; r < b; ++r) { const float v = src[(size_t)r * G + g]; if (v > mx) mx = v; }
            float sum = 0.0f;
            for (i64 r = a; r < b; ++r) {
                float c = src[(size_t)r * G + g] - mx;
                if (scaling) c = c / sq;
                const float e = expf(c);
                out[(size_t)r * G + g] = e;
                sum = (r == a) ? e : sum + e;
            }
            sum = sum + eps;
            for (i64 r = a; r < b; ++r) out[(size_t)r * G + g] = out[(size_t)r * G + g] / sum;
        }
    }
}

/* ------------------------------------------------------------------------ *
 * One MLP layer of core/common_modules/base_modules.py:38-48 in eval mode:
 * y = LeakyReLU_0.2( BN_eval( x @ W^T [+ bias] ) ), or plain Linear when
 * bn_weight == NULL (then 'slope' is ignored).  x: [rows][Cin], W: [Cout][Cin]
 * (nn.Linear layout), y: [rows][Cout].
 * ------------------------------------------------------------------------ */
void orc_linear_bn_act(
    const float *x, i64 rows, int Cin, int Cout, const float *W,
    const float *bias, const float *bn_weight, const float *bn_bias,
    const float *bn_mean, const float *bn_var, float bn_eps, float slope,
    float *y)
{
    for (i64 r = 0; r < rows; ++r) {
        const float *xr = x + (size_t)r * Cin;
        float *yr = y + (size_t)r * Cout;
        for (int o = 0; o < Cout; ++o) {
            const float *w = W + (size_t)o * Cin;
            float acc = 0.0f;
            for (int k = 0; k < Cin; ++k) acc = fmaf(xr[k], w[k], acc);
            if (bias) acc = acc + bias[o];
            if (bn_weight) {
                const float invstd = 1.0f / sqrtf(bn_var[o] + bn_eps);
                acc = (acc - bn_mean[o]) * invstd * bn_weight[o] + bn_bias[o];
                if (acc < 0.0f) acc = acc * slope;
            }
            yr[o] = acc;
        }
    }
}
