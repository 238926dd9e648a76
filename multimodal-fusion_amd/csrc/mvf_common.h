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
// Z-buffer tiles: TILE x TILE pixels of 64-bit keys in LDS.
constexpr int TILE = 64;
constexpr int TILE_LOG = 6;

struct Proj {
    float dist;     // float32 distance to img_xyz
    float x, y;     // float32-valued pixel coordinates (the reference widens them to float64)
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

// camera_projection_cpu for one point (visibility.py:478-538): float32 ops
// separately rounded, the 3x3 rotation as an fmaf chain in k order 0,1,2.
// Returns false when the point is culled.
__device__ __forceinline__ bool project_point(
    float px, float py, float pz, const mvf_view &v, const mvf_vis_params &P,
    const uint8_t *__restrict__ mask, Proj &out)
{
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
    out.dist = d; out.x = x; out.y = y;
    return true;
}

// pinhole_splat_cpu for one point (visibility.py:783-825), float64 island.
__device__ __forceinline__ BBox splat_bbox(
    double x, double y, float dist, float fx, float fy, const mvf_vis_params &P)
{
    const double d = (double)dist;
    const double swell = (1.0 + P.k_swell * exp(-d / P.ln_d_swell)) * P.voxel / d;
    const double wx = swell * (double)fx, wy = swell * (double)fy;
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
