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

enum { CAM_KITTI360_PERSPECTIVE = 0, CAM_SCANNET = 1, CAM_KITTI360_FISHEYE = 2, CAM_S3DIS_EQUIRECTANGULAR = 3 };
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
 * fisheye_projection_cpu  (visibility.py:288-339), 'kitti360_fisheye'.
 * extrinsic: row-major 4x4 float32 camera-to-world; fisheye: xi, k1, k2,
 * gamma1, gamma2, u0, v0.  Typing follows numba (not numpy >= 2): a float32
 * array combined with the float64 literal 1e-4 gives float64, so x, y, z and
 * everything derived from them are float64; p = R.T @ (xyz - T).T and norm_cpu
 * stay float32; norm * p[:, 2] is a float32 product divided in float64.
 * x ** 2 is the exact square (LLVM turns pow(x, 2) into x * x).
 * PARITY UNPINNED at the bit level: the fixtures come from running the
 * reference's file without numba, where numpy keeps float32 throughout, so
 * they pin this function to about 1e-3 pixel only.
 * ------------------------------------------------------------------------ */
static void fisheye_point(
    float px, float py, float pz, const float *M, const float *fisheye,
    double *xo, double *yo, double *zo)
{
    const float e0 = px - M[3], e1 = py - M[7], e2 = pz - M[11];
    const float p0 = fmaf(M[8], e2, fmaf(M[4], e1, M[0] * e0));
    const float p1 = fmaf(M[9], e2, fmaf(M[5], e1, M[1] * e0));
    const float p2 = fmaf(M[10], e2, fmaf(M[6], e1, M[2] * e0));
    const float s0 = p0 * p0, s1 = p1 * p1, s2 = p2 * p2;
    const float s01 = s0 + s1;
    const float ss = s01 + s2;
    const float norm = sqrtf(ss);
    const double xi = (double)fisheye[0], k1 = (double)fisheye[1], k2 = (double)fisheye[2];
    const double g1 = (double)fisheye[3], g2 = (double)fisheye[4];
    const double u0 = (double)fisheye[5], v0 = (double)fisheye[6];
    const double den = (double)norm + 1e-4;
    double x = (double)p0 / den, y = (double)p1 / den;
    const double z = (double)p2 / den;
    const double q = z + xi;
    x = x / q; y = y / q;
    const double r2 = x * x + y * y;
    const double r4 = r2 * r2;
    const double poly = (1.0 + k1 * r2) + k2 * r4;
    *xo = (g1 * poly) * x + u0;
    *yo = (g2 * poly) * y + v0;
    const float np2 = norm * p2;
    *zo = (double)np2 / fabs((double)p2 + 1e-4);
}

void orc_fisheye_projection(
    i64 n, const float *xyz, const float *extrinsic, const float *fisheye,
    double *x, double *y, double *z)
{
    for (i64 i = 0; i < n; ++i)
        fisheye_point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], extrinsic, fisheye,
                      x + i, y + i, z + i);
}

/* a % b of numpy for float64, b > 0 */
static inline double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}

/* ------------------------------------------------------------------------ *
 * equirectangular_projection_cpu  (visibility.py:150-182).  xyz_to_img =
 * xyz - img_xyz, radius = its norm (float32); rotation: row-major 3x3 float32
 * from pose_to_rotation_matrix_cpu (:57-90, computed by the caller).
 * v = xyz_to_img.dot(rotation.T) is the sgemm fmaf chain in k order 0,1,2
 * (measured against numpy on the build host: 0 of 3e6 values differ).
 * arctan2 / arccos of float32 operands are taken correctly rounded (float64
 * evaluation, then rounding): the reference's libm atan2f / acosf may differ
 * from that in the last bit.  Typing follows numba: t / np.pi is float64,
 * (height - 1) * p is a float32 product, then / np.pi in float64.
 * PARITY UNPINNED at the bit level (see orc_fisheye_projection).
 * ------------------------------------------------------------------------ */
void orc_equirectangular_projection(
    i64 n, const float *xyz_to_img, const float *radius, const float *rotation,
    int W, int H, double *x, double *y)
{
    const float *R = rotation;
    for (i64 i = 0; i < n; ++i) {
        const float d0 = xyz_to_img[3 * i], d1 = xyz_to_img[3 * i + 1], d2 = xyz_to_img[3 * i + 2];
        const float v0 = fmaf(R[2], d2, fmaf(R[1], d1, R[0] * d0));
        const float v1 = fmaf(R[5], d2, fmaf(R[4], d1, R[3] * d0));
        const float v2 = fmaf(R[8], d2, fmaf(R[7], d1, R[6] * d0));
        const float t = (float)atan2((double)v1, (double)v0);
        const float c = v2 / radius[i];
        const float p = (float)acos((double)c);
        const double w1 = 1.0 - (double)t / M_PI;
        const double w2 = (double)(W - 1) * w1;
        double wp = py_mod(w2 / 2.0, (double)W);
        const float hp32 = (float)(H - 1) * p;
        double hp = py_mod((double)hp32 / M_PI, (double)H);
        if (isnan(wp)) wp = 0.0;
        if (isnan(hp)) hp = 0.0;
        x[i] = wp; y[i] = hp;
    }
}

/* ------------------------------------------------------------------------ *
 * field_of_view_cpu  (visibility.py:395-435) on float64 coordinates.  z may be
 * NULL (then taken as 1, :525).  Returns the count; out_idx ascending.
 * ------------------------------------------------------------------------ */
i64 orc_field_of_view(
    i64 n, const double *x, const double *y, const double *z, const unsigned char *mask,
    int W, int H, int crop_top, int crop_bottom, i64 *out_idx)
{
    i64 m = 0;
    for (i64 i = 0; i < n; ++i) {
        if (!(0.0 <= x[i])) continue;
        if (!((double)crop_top <= y[i])) continue;
        if (!(x[i] < (double)W)) continue;
        if (!(y[i] < (double)(H - crop_bottom))) continue;
        if (z && !(0.0 < z[i])) continue;
        if (mask) {
            const uint32_t xi = (uint32_t)floor(x[i]), yi = (uint32_t)floor(y[i]);
            if (!mask[(size_t)xi * H + yi]) continue;
        }
        out_idx[m++] = i;
    }
    return m;
}

/* boxes from float64 widths, common tail of the three *_splat_cpu functions
 * (visibility.py:667-702, 807-825, 916-951) */
static void splat_box(
    double x, double y, double wx, double wy, int W, int H, int crop_top, int crop_bottom,
    int32_t *out)
{
    const int32_t x_min = 0, x_max = W, y_min = crop_top, y_max = H - crop_bottom;
    out[0] = clampi(round_f64_to_i32(x - wx / 2), x_min, x_max - 1);
    out[1] = clampi(round_f64_to_i32(x + wx / 2 + 1), x_min + 1, x_max);
    out[2] = clampi(round_f64_to_i32(y - wy / 2), y_min, y_max - 1);
    out[3] = clampi(round_f64_to_i32(y + wy / 2 + 1), y_min + 1, y_max);
}

/* ------------------------------------------------------------------------ *
 * equirectangular_splat_cpu  (visibility.py:630-704), all float64.
 * ------------------------------------------------------------------------ */
void orc_equirectangular_splat(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    int W, int H, int crop_top, int crop_bottom,
    double voxel, double k_swell, double d_swell, int32_t *splat)
{
    const double ln_d = log(d_swell);
    const double b = M_PI / (double)H;
    for (i64 i = 0; i < m; ++i) {
        const double d = (double)dist[i];
        const double aw = (1.0 + k_swell * exp(-d / ln_d)) * voxel / d;
        const double res_y = aw * (double)H / M_PI;
        const double a = aw * (double)W / (2.0 * M_PI);
        const double res_x = a / (sin(b * y_proj[i]) + 0.001);
        splat_box(x_proj[i], y_proj[i], res_x, res_y, W, H, crop_top, crop_bottom, splat + 4 * i);
    }
}

/* ------------------------------------------------------------------------ *
 * fisheye_splat_cpu  (visibility.py:876-953).  xyz: WORLD coordinates of the
 * m splats; the swell uses norm_cpu(xyz), i.e. the distance to the world
 * origin, exactly as written at :900.  z_offset is a float32 array, so
 * swell * voxel / 2 is rounded to float32 before the float32 addition to z.
 * ------------------------------------------------------------------------ */
void orc_fisheye_splat(
    i64 m, const double *x_proj, const double *y_proj, const float *xyz,
    const float *extrinsic, const float *fisheye, int W, int H, int crop_top, int crop_bottom,
    double voxel, double k_swell, double d_swell, int32_t *splat)
{
    const double ln_d = log(d_swell);
    for (i64 i = 0; i < m; ++i) {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        const float s0 = px * px, s1 = py * py, s2 = pz * pz;
        const float s01 = s0 + s1;
        const float ss = s01 + s2;
        const float nw = sqrtf(ss);
        const double swell = 1.0 + k_swell * exp(-(double)nw / ln_d);
        const float lift = (float)(swell * voxel / 2);
        const float pz2 = pz + lift;
        double x, y, z;
        fisheye_point(px, py, pz2, extrinsic, fisheye, &x, &y, &z);
        const double dx = x_proj[i] - x, dy = y_proj[i] - y;
        const double w = 2.0 * sqrt(dx * dx + dy * dy);
        splat_box(x_proj[i], y_proj[i], w, w, W, H, crop_top, crop_bottom, splat + 4 * i);
    }
}

/* ------------------------------------------------------------------------ *
 * visibility_from_splatting_cpu  (visibility.py:1073-1195) from the boxes of
 * any of the *_splat_cpu functions (splat: [m][4], not yet shifted by
 * crop_top).  Sequential z-buffer over a [W][Hc] depth map, strict '<', points
 * visited in ascending index (:1148-1162); exact=1 re-stamps only the splat
 * centres of the seen points, ascending, later overwrites earlier
 * (:1168-1187); output in np.where order of the [W][Hc] map = sorted by x then
 * y (:1190-1195), y expressed in the non-cropped image.  Outputs need room for
 * min(m, W*Hc) (exact) or W*Hc (non exact) entries.  Returns the count, or -1
 * on allocation failure.
 * ------------------------------------------------------------------------ */
i64 orc_zbuffer_from_splat(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    const int32_t *splat, int W, int H, int crop_top, int crop_bottom, int exact,
    i64 *out_idx, i64 *out_x, i64 *out_y)
{
    const int Hc = H - crop_bottom - crop_top;
    const size_t npix = (size_t)W * Hc;
    float *depth = (float *)malloc(sizeof(float) * npix);
    i64 *idmap = (i64 *)malloc(sizeof(i64) * npix);
    if (!depth || !idmap) { free(depth); free(idmap); return -1; }

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
        if (!seen) { free(depth); free(idmap); return -1; }
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
    free(depth); free(idmap);
    return k;
}

/* visibility_from_splatting_cpu for the pinhole cameras (:1121-1125) */
i64 orc_visibility_from_splatting(
    i64 m, const double *x_proj, const double *y_proj, const float *dist,
    float fx, float fy, int W, int H, int crop_top, int crop_bottom,
    double voxel, double k_swell, double d_swell, int exact,
    i64 *out_idx, i64 *out_x, i64 *out_y)
{
    int32_t *splat = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(m > 0 ? m : 1));
    if (!splat) return -1;
    orc_pinhole_splat(m, x_proj, y_proj, dist, fx, fy, W, H, crop_top,
                      crop_bottom, voxel, k_swell, d_swell, splat);
    const i64 k = orc_zbuffer_from_splat(m, x_proj, y_proj, dist, splat, W, H, crop_top,
                                         crop_bottom, exact, out_idx, out_x, out_y);
    free(splat);
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
