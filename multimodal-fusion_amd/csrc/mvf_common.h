// Shared host/device helpers of the mvfusion HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mvfusion.h"

static_assert(sizeof(mvf_view) == 128, "mvf_view must stay 128 bytes");

namespace mvf {

typedef long long i64;
typedef unsigned long long u64;

// ---- error reporting ------------------------------------------------------
void set_error(const char *fmt, ...);

#define MVF_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            mvf::set_error(__VA_ARGS__);                         \
            return MVF_E_INVALID;                                \
        }                                                        \
    } while (0)

#define MVF_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) {                                              \
            mvf::set_error("%s failed: %s (%s:%d)", #call,                   \
                           hipGetErrorString(e_), __FILE__, __LINE__);       \
            return MVF_E_LAUNCH;                                             \
        }                                                                    \
    } while (0)

#define MVF_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            mvf::set_error("kernel launch failed: %s (%s:%d)",               \
                           hipGetErrorString(e_), __FILE__, __LINE__);       \
            return MVF_E_LAUNCH;                                             \
        }                                                                    \
    } while (0)

// Stage timing hooks (no-ops unless mvf_profile_enable(1) was called on this thread).
void prof_begin(hipStream_t st);
void prof_mark(const char *name, hipStream_t st);

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char *base;
    size_t size, off;
    Arena(void *p, size_t n) : base((char *)p), size(n), off(0) {}
    template <typename T> T *take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        T *r = (T *)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= size && (base != nullptr || off == 0); }
};

static inline unsigned cdiv(i64 a, i64 b) { return (unsigned)((a + b - 1) / b); }

// ---- projection math shared by every mapping kernel -------------------------
// Z-buffer tiles: TILE_X x TILE_Y pixels of 64-bit keys in LDS, one workgroup of
// TILE_X * TILE_Y / 8 threads per tile.
#ifndef MVF_TILE_Y_LOG
#define MVF_TILE_Y_LOG 6
#endif
constexpr int TILE_X_LOG = 6, TILE_Y_LOG = MVF_TILE_Y_LOG;
constexpr int TILE_X = 1 << TILE_X_LOG, TILE_Y = 1 << TILE_Y_LOG;

struct Proj {
    float dist;     // float32 distance to img_xyz
    double x, y;    // pixel coordinates as the reference returns them (float64)
};

struct BBox {
    int xa, xb, ya, yb;  // [xa, xb) x [ya, yb), y already relative to crop_top
};

// float64 -> rint -> float32 -> int32 with the x86 out-of-range result (INT_MIN)
__device__ __forceinline__ int round_f64_to_i32(double v)
{
    const float f = (float)rint(v);
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int)f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

// p = R^T (xyz - T) of the 'kitti360_*' cameras (visibility.py:240-242,305-308):
// float32, the 3x3 product as an fmaf chain in k order 0,1,2 (what the sgemm
// of the reference computes).
__device__ __forceinline__ float kitti_row(const float *M, int c, float e0, float e1, float e2)
{
    return __fmaf_rn(M[8 + c], e2, __fmaf_rn(M[4 + c], e1, __fmul_rn(M[c], e0)));
}

// fisheye_projection_cpu for one point (visibility.py:288-339).  Typing follows
// numba: the float64 literal 1e-4 promotes norm + 1e-4 and everything computed
// from it to float64, products of two float32 arrays stay float32.
__device__ __forceinline__ void fisheye_project(
    float px, float py, float pz, const mvf_view &v, double &x, double &y, double &z)
{
    const float *M = v.cam_to_world;
    const float e0 = __fsub_rn(px, M[3]), e1 = __fsub_rn(py, M[7]), e2 = __fsub_rn(pz, M[11]);
    const float p0 = kitti_row(M, 0, e0, e1, e2), p1 = kitti_row(M, 1, e0, e1, e2),
                p2 = kitti_row(M, 2, e0, e1, e2);
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(p0, p0), __fmul_rn(p1, p1)), __fmul_rn(p2, p2)));
    const double den = (double)nrm + 1e-4;
    double xs = (double)p0 / den, ys = (double)p1 / den;
    const double q = (double)p2 / den + (double)M[12];                 // z + xi
    xs = xs / q; ys = ys / q;
    const double r2 = xs * xs + ys * ys, r4 = r2 * r2;
    const double s = (1.0 + (double)M[13] * r2) + (double)M[14] * r4;  // 1 + k1 r2 + k2 r4
    x = ((double)v.fx * s) * xs + (double)v.cx;                        // gamma1 (..) x + u0
    y = ((double)v.fy * s) * ys + (double)v.cy;
    z = (double)__fmul_rn(nrm, p2) / fabs((double)p2 + 1e-4);
}

// Width (= height) in pixels of a fisheye splat (fisheye_splat_cpu,
// visibility.py:900-914): distance between the projection (x, y) of the point
// and that of the point lifted by swell * voxel / 2 along the world z axis.
// The swell uses the norm of the WORLD coordinates, as the reference does.
__device__ __forceinline__ double fisheye_splat_width(
    float px, float py, float pz, double x, double y, const mvf_view &v, const mvf_vis_params &P)
{
    const float nw = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz)));
    const double swell = 1.0 + P.k_swell * exp(-(double)nw / P.ln_d_swell);
    const float lift = (float)(swell * P.voxel / 2.0);
    double x2, y2, z2;
    fisheye_project(px, py, __fadd_rn(pz, lift), v, x2, y2, z2);
    const double dx = x - x2, dy = y - y2;
    return 2.0 * sqrt(dx * dx + dy * dy);
}

// a % b of numpy / numba for float64 (result takes the sign of b > 0)
__device__ __forceinline__ double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}

constexpr double MVF_PI = 3.141592653589793;

// equirectangular_projection_cpu for one point (visibility.py:150-182): d =
// xyz - img_xyz and radius = its norm, both float32.  arctan2 / arccos of
// float32 operands are taken correctly rounded (evaluated in float64, then
// rounded); t / pi and everything after it is float64 as numba types it, while
// (height - 1) * p stays a float32 product.
__device__ __forceinline__ void equirect_project(
    float dx, float dy, float dz, float radius, const mvf_view &v, const mvf_vis_params &P,
    double &x, double &y)
{
    const float *R = v.cam_to_world;      // row-major 3x3
    const float v0 = __fmaf_rn(R[2], dz, __fmaf_rn(R[1], dy, __fmul_rn(R[0], dx)));
    const float v1 = __fmaf_rn(R[5], dz, __fmaf_rn(R[4], dy, __fmul_rn(R[3], dx)));
    const float v2 = __fmaf_rn(R[8], dz, __fmaf_rn(R[7], dy, __fmul_rn(R[6], dx)));
    const float t = (float)atan2((double)v1, (double)v0);
    const float p = (float)acos((double)__fdiv_rn(v2, radius));
    x = py_mod((double)(P.W - 1) * (1.0 - (double)t / MVF_PI) / 2.0, (double)P.W);
    y = py_mod((double)__fmul_rn((float)(P.H - 1), p) / MVF_PI, (double)P.H);
    if (x != x) x = 0.0;
    if (y != y) y = 0.0;
}

// camera_projection_cpu for one point (visibility.py:478-538): float32 ops
// separately rounded, the 3x3 rotation as an fmaf chain in k order 0,1,2.
// Returns false when the point is culled.  ANY = false compiles the pinhole
// cameras only (the fused all-views kernels).
template <bool ANY>
__device__ __forceinline__ bool project_point(
    float px, float py, float pz, const mvf_view &v, const mvf_vis_params &P,
    const uint8_t *__restrict__ mask, Proj &out)
{
    if (ANY && P.camera >= MVF_CAM_KITTI360_FISHEYE) {
        const float dx = __fsub_rn(px, v.img_xyz[0]);
        const float dy = __fsub_rn(py, v.img_xyz[1]);
        const float dz = __fsub_rn(pz, v.img_xyz[2]);
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const float d = sqrtf(s);
        if (!(P.r_min < (double)d && (double)d < P.r_max)) return false;
        double x, y, z = 1.0;
        if (P.camera == MVF_CAM_KITTI360_FISHEYE) fisheye_project(px, py, pz, v, x, y, z);
        else equirect_project(dx, dy, dz, d, v, P, x, y);
        // field_of_view_cpu (visibility.py:395-435) on the float64 coordinates
        if (!(0.0 <= x && (double)P.crop_top <= y && x < (double)P.W &&
              y < (double)(P.H - P.crop_bottom) && 0.0 < z)) return false;
        if (mask) {
            const unsigned xi = (unsigned)floor(x), yi = (unsigned)floor(y);
            if (!mask[(size_t)xi * P.H + yi]) return false;
        }
        out.dist = d; out.x = x; out.y = y;
        return true;
    }
    // A point survives when it passes the range test AND every field-of-view
    // test, so the tests may run in any order: the cheap ones (depth sign, then
    // the image bounds) come first and the square root / float64 range compare
    // only for what is left.  Every value is computed exactly as in the
    // reference's order of operations.
    const float *M = v.cam_to_world;
    float p0, p1, p2;
    if (P.camera == MVF_CAM_KITTI360_PERSPECTIVE) {
        const float e0 = __fsub_rn(px, M[3]), e1 = __fsub_rn(py, M[7]), e2 = __fsub_rn(pz, M[11]);
        p2 = __fmaf_rn(M[10], e2, __fmaf_rn(M[6], e1, __fmul_rn(M[2], e0)));
        if (!(0.0f < p2)) return false;
        p0 = __fmaf_rn(M[8], e2, __fmaf_rn(M[4], e1, __fmul_rn(M[0], e0)));
        p1 = __fmaf_rn(M[9], e2, __fmaf_rn(M[5], e1, __fmul_rn(M[1], e0)));
    } else {
        p2 = __fadd_rn(__fmaf_rn(M[10], pz, __fmaf_rn(M[9], py, __fmul_rn(M[8], px))), M[11]);
        if (!(0.0f < p2)) return false;
        p0 = __fadd_rn(__fmaf_rn(M[2], pz, __fmaf_rn(M[1], py, __fmul_rn(M[0], px))), M[3]);
        p1 = __fadd_rn(__fmaf_rn(M[6], pz, __fmaf_rn(M[5], py, __fmul_rn(M[4], px))), M[7]);
    }
    // field_of_view_cpu (visibility.py:395-435); float32 values compare exactly
    // like their float64 widenings
    const float x = __fadd_rn(__fdiv_rn(__fmul_rn(p0, v.fx), p2), v.cx);
    if (!(0.0f <= x)) return false;
    if (!(x < (float)P.W)) return false;
    const float y = __fadd_rn(__fdiv_rn(__fmul_rn(p1, v.fy), p2), v.cy);
    if (!((float)P.crop_top <= y)) return false;
    if (!(y < (float)(P.H - P.crop_bottom))) return false;
    const float dx = __fsub_rn(px, v.img_xyz[0]);
    const float dy = __fsub_rn(py, v.img_xyz[1]);
    const float dz = __fsub_rn(pz, v.img_xyz[2]);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float d = sqrtf(s);
    if (!(P.r_min < (double)d && (double)d < P.r_max)) return false;
    if (mask) {
        const unsigned xi = (unsigned)floorf(x), yi = (unsigned)floorf(y);
        if (!mask[(size_t)xi * P.H + yi]) return false;
    }
    out.dist = d; out.x = (double)x; out.y = (double)y;
    return true;
}

// Splat box of one point (float64 island).  (sx, sy) is what the camera's
// splat needs beyond (x, y, dist): pinhole_splat_cpu (visibility.py:783-825)
// scales the swell by fx, fy; fisheye_splat_cpu (:914-951) takes the box width
// computed by fisheye_splat_width; equirectangular_splat_cpu (:651-702) derives
// both widths from the image size and the latitude of y.
template <bool ANY>
__device__ __forceinline__ BBox splat_bbox(
    double x, double y, float dist, double sx, double sy, const mvf_vis_params &P)
{
    double wx, wy;
    if (ANY && P.camera == MVF_CAM_KITTI360_FISHEYE) {
        wx = sx; wy = sy;
    } else {
        const double d = (double)dist;
        const double swell = (1.0 + P.k_swell * exp(-d / P.ln_d_swell)) * P.voxel / d;
        if (ANY && P.camera == MVF_CAM_S3DIS_EQUIRECTANGULAR) {
            wy = swell * (double)P.H / MVF_PI;
            const double a = swell * (double)P.W / (2.0 * MVF_PI);
            const double b = MVF_PI / (double)P.H;
            wx = a / (sin(b * y) + 0.001);
        } else {
            wx = swell * sx; wy = swell * sy;
        }
    }
    const int y_min = P.crop_top, y_max = P.H - P.crop_bottom;
    BBox b;
    b.xa = clampi(round_f64_to_i32(x - wx / 2), 0, P.W - 1);
    b.xb = clampi(round_f64_to_i32(x + wx / 2 + 1), 1, P.W);
    b.ya = clampi(round_f64_to_i32(y - wy / 2), y_min, y_max - 1) - P.crop_top;
    b.yb = clampi(round_f64_to_i32(y + wy / 2 + 1), y_min + 1, y_max) - P.crop_top;
    return b;
}

// One (splat, tile) pair handed to the z-buffer tile kernel.
struct __attribute__((aligned(16))) TileEntry {
    u64 key;                  // float bits of dist << 32 | local point index
    uint16_t xa, xb, ya, yb;  // bbox in cropped image coordinates
};
static_assert(sizeof(TileEntry) == 16, "TileEntry must be 16 bytes");

__device__ __forceinline__ u64 depth_key(float dist, unsigned idx)
{
    return ((u64)__float_as_uint(dist) << 32) | (u64)idx;
}

}  // namespace mvf
