// Point -> pixel mapping kernels: camera projection, splat z-buffer visibility,
// depth-map visibility, mapping features, and the fused all-views MapImages
// path.  gfx950 only.
//
// Z-buffer design.  The image is cut into tiles of TILE_X x TILE_Y pixels.
// Splats are binned to the tiles they overlap (count -> scan -> fill, order
// inside a bin is irrelevant), then one workgroup per tile resolves its tile in
// LDS: a 64-bit key  float_bits(dist) << 32 | local_index  per pixel and per
// cell of 4 vertically adjacent pixels; a splat does an LDS atomicMin on every
// cell it covers entirely and on the pixels of its remaining rows, and the
// depth of a pixel is the smaller of its own key and its cell's key.  dist > 0
// makes the float bit pattern order preserving, so the minimum key is exactly
// the reference's sequential rule "strictly closer wins, first index wins a
// tie" (visibility.py:1148-1162), independent of execution order.  The depth
// buffer never touches HBM.
#include <mutex>

#include "mvf_common.h"
#include "mvf_scan.h"

namespace mvf {

constexpr u64 ZB_EMPTY = ~0ull;

// ------------------------------------------------------------------------- //
// camera_projection                                                           //
// ------------------------------------------------------------------------- //
struct ProjPred {
    const float *xyz; mvf_view view; mvf_vis_params P; const uint8_t *mask;
    __device__ bool operator()(i64 i) const {
        Proj pr;
        return project_point<true>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], view, P, mask, pr);
    }
};
struct ProjEmit {
    const float *xyz; mvf_view view; mvf_vis_params P; const uint8_t *mask;
    i64 *indices; float *dist; double *x; double *y;
    __device__ void operator()(i64 i, i64 pos) const {
        Proj pr;
        project_point<true>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], view, P, mask, pr);
        indices[pos] = i; dist[pos] = pr.dist; x[pos] = pr.x; y[pos] = pr.y;
    }
};

// ------------------------------------------------------------------------- //
// splat sources: where a kernel gets (x, y, dist, sx, sy) of splat i, view v  //
// ------------------------------------------------------------------------- //
// (sx, sy) is the camera-specific input of splat_bbox.
struct ArraySrc {                      // visibility_from_splatting(x_proj, y_proj, dist, xyz)
    const double *x, *y; const float *dist; const float *xyz; mvf_view view; i64 m;
    static constexpr bool ANY = true;
    __device__ i64 count() const { return m; }
    __device__ int2 views_of(i64, int) const { return make_int2(0, 1); }
    __device__ bool get(i64 i, int, const mvf_vis_params &P, double &ox, double &oy, float &od,
                        double &sx, double &sy) const {
        ox = x[i]; oy = y[i]; od = dist[i];
        if (P.camera == MVF_CAM_KITTI360_FISHEYE)
            sx = sy = fisheye_splat_width(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], ox, oy, view, P);
        else { sx = (double)view.fx; sy = (double)view.fy; }
        return true;
    }
};

// fused MapImages: project xyz[i] into view v.  ANY_ = false: pinhole cameras
// only (no trace of the other camera models in the kernels); true: any camera.
template <bool ANY_>
struct CloudSrcT {
    const float *xyz; const mvf_view *views; const uint8_t *mask; i64 n; int cylinder;
    // Batch of samples in one call: window[i] = {first view, number of views} of
    // point position i (the views of its own sample); null = every point meets
    // all V views.  Kernels walk a point's views by their index inside the window.
    const int2 *window;
    static constexpr bool ANY = ANY_;
    __device__ i64 count() const { return n; }
    __device__ int2 views_of(i64 i, int V) const { return window ? window[i] : make_int2(0, V); }
    __device__ bool get(i64 i, int v, const mvf_vis_params &P, double &ox, double &oy, float &od,
                        double &sx, double &sy) const {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        const mvf_view &view = views[v];
        Proj pr;
        if (!project_point<ANY_>(px, py, pz, view, P, mask, pr)) return false;
        // radius sampling of MapImages (image.py:242-245; transforms.py:332,386):
        // float64 squared distance to the view centre <= r_max^2 (tested last: it
        // only removes points, and few reach it)
        const double ex = (double)px - view.centre[0], ey = (double)py - view.centre[1];
        double d2 = ex * ex + ey * ey;
        if (!cylinder) { const double ez = (double)pz - view.centre[2]; d2 = d2 + ez * ez; }
        if (!(d2 <= P.r_max * P.r_max)) return false;
        ox = pr.x; oy = pr.y; od = pr.dist;
        if (ANY_ && P.camera == MVF_CAM_KITTI360_FISHEYE) sx = sy = fisheye_splat_width(px, py, pz, ox, oy, view, P);
        else { sx = (double)view.fx; sy = (double)view.fy; }
        return true;
    }
};
using CloudSrc = CloudSrcT<false>;
using CloudSrcAny = CloudSrcT<true>;

// ------------------------------------------------------------------------- //
// binning: count, fill                                                        //
// ------------------------------------------------------------------------- //
// Consecutive points are close in space (the sampler's order), so the 256
// points of a block touch few tiles.  Tile counters are therefore first
// accumulated in an LDS histogram of the view's tiles and flushed with one
// global atomic per touched tile: global atomics on the hot tile counters
// otherwise serialise at the memory side.  The histogram covers the block's
// range of touched tile indices (BIN_LDS counters, 8 KB, so that LDS does not
// limit the number of resident blocks); a block whose range is wider uses
// global atomics directly.
//
// Inside a tile the entries are grouped by decreasing clipped pixel count
// (NCLS size classes, a factor 2 apart): the z-buffer kernel deals consecutive
// entries to lanes that share an instruction stream, and entries of similar
// size keep those lanes busy for the same number of steps.  Counters are
// therefore indexed [view][tile][class].
constexpr int NCLS = 8;
// Behind the counters of a call: the number of non-empty tiles, then the ticket
// counters of the z-buffer kernel, one per 128 bytes (atomics on one address
// serialise at the memory side, so the tickets are spread over ZB_QUEUES
// counters).
constexpr int ZB_QUEUES = 32;
constexpr int ZB_QUEUE_STRIDE = 32;
constexpr int BIN_TAIL = ZB_QUEUE_STRIDE * (ZB_QUEUES + 1);
// a spare counter of the tail (cleared with it): entries handed out in the verdict buffer of
// the plain-splat pass
constexpr int NE_CURSOR_SLOT = 8;
#ifndef MVF_BIN_LDS
#define MVF_BIN_LDS 2048
#endif
constexpr int BIN_LDS = MVF_BIN_LDS;         // LDS histogram entries per block, fused path (points in spatial order)
constexpr int BIN_LDS_WIDE = 12288;          // single-view entry point: points in any order (48 KB)

// Raster cost of a (splat, tile) pair: atomic instructions a row of 16 lanes
// needs for the clipped box of cw x ch pixels.  The lanes cover 2^lg >= ch rows
// (at most 16) of 16 >> lg neighbouring columns at a time; boxes taller than
// 16 rows take one pass per band of 16 rows.
__device__ __forceinline__ unsigned raster_lg(unsigned ch)
{
    const unsigned hm = ch < 16u ? ch : 16u;
    return hm <= 1u ? 0u : 32u - (unsigned)__clz(hm - 1u);
}

// Raster steps of a rectangle of w x h cells (h <= 16) for a row of 16 lanes
// shaped 2^lg >= h rows by 16 >> lg columns.
__device__ __forceinline__ unsigned rect_steps(unsigned w, unsigned h)
{
    if (h == 0u) return 0u;
    const unsigned sh = 4u - raster_lg(h);
    return (w + (1u << sh) - 1u) >> sh;
}

// The z-buffer of a tile has two levels: pixels, and cells of ZB_CELL
// vertically adjacent pixels (the cell grid is aligned to the image rows).  A
// box writes its key once per cell it covers entirely and once per pixel of the
// rows above and below its cells; the depth of a pixel is the smaller of its own
// key and its cell's key.  Rows of the clipped box [ya, yb) that are covered by
// whole cells: [cell_lo, cell_hi) (empty when cell_hi <= cell_lo).
constexpr int ZB_CELL_LOG = 2;
constexpr int ZB_CELL = 1 << ZB_CELL_LOG;
__device__ __forceinline__ void cell_rows(int ya, int yb, int &cell_lo, int &cell_hi)
{
    cell_lo = (ya + ZB_CELL - 1) & ~(ZB_CELL - 1);
    cell_hi = yb & ~(ZB_CELL - 1);
}

// Size class of a (splat, tile) pair: 0 = most expensive, classes a factor 2 apart.
__device__ __forceinline__ int size_class(const BBox &b, int tx, int ty)
{
    const int x0 = tx << TILE_X_LOG, y0 = ty << TILE_Y_LOG;
    const unsigned cw = (b.xb < x0 + TILE_X ? b.xb : x0 + TILE_X) - (b.xa > x0 ? b.xa : x0);
    const int ya = b.ya > y0 ? b.ya : y0, yb = b.yb < y0 + TILE_Y ? b.yb : y0 + TILE_Y;
    int lo, hi;
    cell_rows(ya, yb, lo, hi);
    unsigned cost;                                                 // 1 .. 64 + 2 * 16
    if (hi > lo) cost = rect_steps(cw, (unsigned)(hi - lo) >> ZB_CELL_LOG) + rect_steps(cw, (unsigned)((lo - ya) + (yb - hi)));
    else cost = rect_steps(cw, (unsigned)(yb - ya));
    const int l = cost <= 1u ? 0 : 32 - __clz(cost - 1u);         // ceil(log2(cost)), 0 .. 7
    return (NCLS - 1) - (l > NCLS - 1 ? NCLS - 1 : l);
}

// Rank of the thread's flag inside the block, number of flags, and the range
// of tile indices [lo, hi) touched by the flagged threads' boxes (hi <= lo when
// there is none), with a single barrier.  The points of a block are neighbours
// in space, so the range is a small part of the view's tiles and the LDS
// histogram only needs clearing and flushing there.
// `view` / `one_view`: the view of the flagged threads when they all agree (a
// block of a batched call may hold points of two samples, whose views differ).
struct BlockBins { unsigned rank, total; int lo, hi, view; bool one_view; };

// VIEWS = false: every thread of the block is on view v (no reduction over it).
template <bool VIEWS>
__device__ __forceinline__ BlockBins block_bins(bool have, int t_first, int t_last, int v)
{
    __shared__ int s_cnt[4], s_mn[4], s_mx[4], s_vmn[4], s_vmx[4];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 m = __ballot(have);
    int mn = have ? t_first : 0x7fffffff, mx = have ? t_last : -1;
    int vmn = have ? v : 0x7fffffff, vmx = have ? v : -1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        if (VIEWS) {
            const int ovmn = __shfl_xor(vmn, d, 64), ovmx = __shfl_xor(vmx, d, 64);
            vmn = ovmn < vmn ? ovmn : vmn;
            vmx = ovmx > vmx ? ovmx : vmx;
        }
    }
    if (lane == 0) {
        s_cnt[wave] = __popcll(m); s_mn[wave] = mn; s_mx[wave] = mx;
        if (VIEWS) { s_vmn[wave] = vmn; s_vmx[wave] = vmx; }
    }
    __syncthreads();
    BlockBins r;
    r.rank = __popcll(m & ((1ull << lane) - 1ull));
    r.total = 0; r.lo = 0x7fffffff; r.hi = -1;
    int vlo = 0x7fffffff, vhi = -1;
#pragma unroll
    for (unsigned w = 0; w < 4; ++w) {
        const int c = s_cnt[w];
        if (w < wave) r.rank += c;
        r.total += c;
        r.lo = s_mn[w] < r.lo ? s_mn[w] : r.lo;
        r.hi = s_mx[w] > r.hi ? s_mx[w] : r.hi;
        if (VIEWS) {
            vlo = s_vmn[w] < vlo ? s_vmn[w] : vlo;
            vhi = s_vmx[w] > vhi ? s_vmx[w] : vhi;
        }
    }
    r.hi += 1;
    r.view = VIEWS ? vlo : v;
    r.one_view = VIEWS ? vlo == vhi : true;
    return r;
}

// WIN: the call is a batch of samples (src.window set): a thread's view is the
// blockIdx.y-th of its point's window, and a block may meet two views.
template <class Src, bool ZERO_SEEN, int NH, bool WIN = false>
__global__ __launch_bounds__(256) void bin_count_kernel(
    Src src, mvf_vis_params P, int tiles_y, int tiles_per_view,
    unsigned *__restrict__ tile_count, uint8_t *__restrict__ seen, i64 seen_stride,
    i64 *__restrict__ zero_counters, int *__restrict__ zero_view_seen, int V,
    TileEntry *__restrict__ staged, unsigned *__restrict__ staged_count)
{
    __shared__ unsigned hist[NH];
    const int tid = threadIdx.x;
    const i64 i = (i64)blockIdx.x * 256 + tid;
    const int lv = blockIdx.y;             // index of the view among the views of the thread's point
    // the first block also clears the caller's small output arrays
    if (blockIdx.x == 0 && lv == 0) {
        if (zero_counters && tid < 8) zero_counters[tid] = 0;
        if (zero_view_seen) for (int k = tid; k < V; k += 256) zero_view_seen[k] = 0;
    }
    const bool live = i < src.count();
    const int2 win = WIN && live ? src.views_of(i, V) : make_int2(0, WIN ? 0 : 0x7fffffff);
    const int v = win.x + lv;
    if (ZERO_SEEN && live && lv == 0) {
        // flags of a point over all views: seen_stride bytes, a multiple of 16
        uint4 *row = (uint4 *)(seen + (size_t)i * seen_stride);
        for (int k = 0; k < (int)(seen_stride >> 4); ++k) row[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    double x, y, sx, sy; float d = 0.f;
    const bool have = live && lv < win.y && src.get(i, v, P, x, y, d, sx, sy);
    BBox b{0, 0, 0, 0};
    if (have) b = splat_bbox<Src::ANY>(x, y, d, sx, sy, P);
    const int tx0 = b.xa >> TILE_X_LOG, tx1 = (b.xb - 1) >> TILE_X_LOG;
    const int ty0 = b.ya >> TILE_Y_LOG, ty1 = (b.yb - 1) >> TILE_Y_LOG;
    const int ncnt = tiles_per_view * NCLS;
    const BlockBins bb = block_bins<WIN>(have, tx0 * tiles_y + ty0, tx1 * tiles_y + ty1, v);
    // the LDS histogram holds the tiles of one view
    unsigned *tc = tile_count + (size_t)(bb.one_view ? bb.view : v) * ncnt;
    {
        // the boxes are kept for the fill pass, packed per block
        const size_t blk = (size_t)lv * gridDim.x + blockIdx.x;
        if (tid == 0) staged_count[blk] = bb.total;
        if (have) {
            TileEntry e;
            e.key = depth_key(d, (unsigned)i);
            e.xa = (uint16_t)b.xa; e.xb = (uint16_t)b.xb; e.ya = (uint16_t)b.ya; e.yb = (uint16_t)b.yb;
            staged[blk * 256 + bb.rank] = e;
        }
    }
    if (bb.total == 0) return;
    const int lo = bb.lo * NCLS, hi = bb.hi * NCLS;
    if (bb.one_view && hi - lo <= NH) {
        for (int t = tid; t < hi - lo; t += 256) hist[t] = 0;
        __syncthreads();
        if (have)
            for (int tx = tx0; tx <= tx1; ++tx)
                for (int ty = ty0; ty <= ty1; ++ty)
                    atomicAdd(&hist[(tx * tiles_y + ty) * NCLS + size_class(b, tx, ty) - lo], 1u);
        __syncthreads();
        for (int t = tid; t < hi - lo; t += 256) {
            const unsigned c = hist[t];
            if (c) atomicAdd(&tc[lo + t], c);
        }
    } else if (have) {
        for (int tx = tx0; tx <= tx1; ++tx)
            for (int ty = ty0; ty <= ty1; ++ty)
                atomicAdd(&tc[(tx * tiles_y + ty) * NCLS + size_class(b, tx, ty)], 1u);
    }
}

template <int NH, bool WIN = false>
__global__ __launch_bounds__(256) void bin_fill_kernel(
    const TileEntry *__restrict__ staged, const unsigned *__restrict__ staged_count,
    int tiles_y, int tiles_per_view,
    const unsigned *__restrict__ tile_off, unsigned *__restrict__ tile_cur,
    TileEntry *__restrict__ entries, i64 pair_capacity, i64 *__restrict__ overflow,
    const int2 *__restrict__ window)
{
    __shared__ unsigned hist[NH];        // per (tile, class): count, then absolute write cursor
    const int tid = threadIdx.x;
    const int lv = blockIdx.y;
    const size_t blk = (size_t)lv * gridDim.x + blockIdx.x;
    const unsigned n = staged_count[blk];
    if (n == 0) return;
    const bool have = (unsigned)tid < n;
    TileEntry e;
    e.key = 0; e.xa = e.xb = e.ya = e.yb = 0;
    if (have) e = staged[blk * 256 + tid];
    // the view of the entry: the lv-th of its point's views
    const int v = (WIN && have ? window[(unsigned)e.key].x : 0) + lv;
    const BBox b{(int)e.xa, (int)e.xb, (int)e.ya, (int)e.yb};
    const int tx0 = b.xa >> TILE_X_LOG, tx1 = (b.xb - 1) >> TILE_X_LOG;
    const int ty0 = b.ya >> TILE_Y_LOG, ty1 = (b.yb - 1) >> TILE_Y_LOG;
    const int ncnt = tiles_per_view * NCLS;
    const BlockBins bb = block_bins<WIN>(have, tx0 * tiles_y + ty0, tx1 * tiles_y + ty1, v);
    const size_t cbase = (size_t)(bb.one_view ? bb.view : v) * ncnt;
    const int lo = bb.lo * NCLS, hi = bb.hi * NCLS;
    if (bb.one_view && hi - lo <= NH) {
        for (int t = tid; t < hi - lo; t += 256) hist[t] = 0;
        __syncthreads();
        if (have)
            for (int tx = tx0; tx <= tx1; ++tx)
                for (int ty = ty0; ty <= ty1; ++ty)
                    atomicAdd(&hist[(tx * tiles_y + ty) * NCLS + size_class(b, tx, ty) - lo], 1u);
        __syncthreads();
        // one global reservation per touched counter; hist becomes the write cursor
        for (int t = tid; t < hi - lo; t += 256) {
            const unsigned c = hist[t];
            if (c) hist[t] = tile_off[cbase + lo + t] + atomicAdd(&tile_cur[cbase + lo + t], c);
        }
        __syncthreads();
        if (have)
            for (int tx = tx0; tx <= tx1; ++tx)
                for (int ty = ty0; ty <= ty1; ++ty) {
                    const int t = (tx * tiles_y + ty) * NCLS + size_class(b, tx, ty) - lo;
                    const i64 pos = (i64)atomicAdd(&hist[t], 1u);
                    if (pos < pair_capacity) entries[pos] = e;
                    else *overflow = 1;
                }
    } else if (have) {
        for (int tx = tx0; tx <= tx1; ++tx)
            for (int ty = ty0; ty <= ty1; ++ty) {
                const size_t t = cbase + (tx * tiles_y + ty) * NCLS + size_class(b, tx, ty);
                const i64 pos = (i64)tile_off[t] + atomicAdd(&tile_cur[t], 1u);
                if (pos < pair_capacity) entries[pos] = e;
                else *overflow = 1;
            }
    }
}

// ------------------------------------------------------------------------- //
// z-buffer of one tile in LDS                                                 //
// ------------------------------------------------------------------------- //
// LDS image: pixel level zb[x][y], 8-byte keys, y fastest, followed by the cell
// level zc[x][y / ZB_CELL].
//
// An LDS atomicMin costs the same whatever the number of active lanes
// (measured on gfx950: about 6.5 cycles per wave-instruction per CU for 64-bit
// keys, as much as a plain 64-bit LDS write), so the kernel is built to issue
// as few of them as possible and to keep LDS waits out of the loop.  A box is
// cut into two rectangles: the cells it covers entirely (cell level) and the
// pixel rows above and below them (pixel level, swept as one rectangle with a
// gap); a box without a whole cell is one pixel rectangle.  That cuts the
// atomics of the mean splat by 2.6 against one atomic per pixel.  The 512 threads form 32 "quarters" (rows of
// 16 lanes).  The bin is consumed 512 entries at a time, one entry per lane.
// The four quarters of a wave walk the bin four entries at a time in one
// instruction stream (the bin is ordered by raster cost, so they get entries
// of similar size).  A quarter takes its entry from lane J of its row with DPP
// row broadcasts (no LDS traffic) and its 16 lanes sweep each rectangle as
// 2^lg >= h rows by 16 >> lg columns per step.
template <int J> __device__ __forceinline__ unsigned row_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x150 + J, 0xf, 0xf, false);
}

// Column pitch of the pixel level: 64 + 9 keys.  With 32 8-byte slots per bank
// row, neighbouring columns start 9 slots apart, which keeps the 16 lanes of a
// row on distinct slots for the lane shapes 1x16, 4x4, 8x2, 16x1 (2x8: one pair
// shares a slot).  Cell level: 16 cells per column, pitch 26.
constexpr unsigned ZB_PITCH = TILE_Y + 9;
constexpr unsigned ZB_PITCH_BYTES = ZB_PITCH * 8;
constexpr unsigned ZC_ROWS = TILE_Y >> ZB_CELL_LOG;
constexpr unsigned ZC_PITCH = ZC_ROWS + (ZC_ROWS == 16 ? 10 : 2);
constexpr unsigned ZC_PITCH_BYTES = ZC_PITCH * 8;
constexpr unsigned ZC_BASE_BYTES = TILE_X * ZB_PITCH_BYTES;            // cell level behind the pixel level
constexpr unsigned ZB_KEYS = TILE_X * (ZB_PITCH + ZC_PITCH);
static_assert(ZB_KEYS * 8 < 65536, "byte offsets of the LDS image are staged in 16 bits");
static_assert(TILE_Y * 8 + 8 <= ZB_PITCH_BYTES && ZC_ROWS * 8 + 8 <= ZC_PITCH_BYTES, "row offsets stay below the column pitch");

// One rectangle of a staged entry: geom = byte offset in the LDS image of its
// first cell | byte offset one past its last column << 16; shape = h | lg << 8 |
// step << 16 with 2^lg = rows covered by the 16 lanes and step = byte distance
// of 16 >> lg columns.  h = 0: no rectangle.  The pixel rows above and below the
// cells of a box form ONE rectangle of h = top + bottom rows: its rows from
// `top` on lie `gap` bytes further down (split = top | gap << 8).
template <unsigned PITCH_BYTES, bool SPLIT>
__device__ __forceinline__ void raster_rect(char *zb, u64 key, unsigned geom, unsigned shape, unsigned split, unsigned ly)
{
    const unsigned h = shape & 0xffu;
    const unsigned lg = (shape >> 8) & 0xffu;
    const unsigned xo = ly >> lg, yo = ly - (xo << lg);
    if (yo < h) {
        const unsigned step = shape >> 16, end = geom >> 16;
        unsigned a = (geom & 0xffffu) + xo * PITCH_BYTES + (yo << 3);
        if (SPLIT) a += yo >= (split & 0xffu) ? split >> 8 : 0u;
        // (row offsets stay below the column pitch, so the last column ends before `end`)
        for (; a < end; a += step) atomicMin((u64 *)(zb + a), key);
    }
}

struct StagedEntry { unsigned klo, khi, geom[2], shape[2], split; };      // [0]: cells, [1]: pixel rows

template <int J>
__device__ __forceinline__ void raster_entry(char *zb, const StagedEntry &mine, unsigned ly)
{
    const u64 key = ((u64)row_bcast<J>(mine.khi) << 32) | row_bcast<J>(mine.klo);
    raster_rect<ZC_PITCH_BYTES, false>(zb, key, row_bcast<J>(mine.geom[0]), row_bcast<J>(mine.shape[0]), 0u, ly);
    raster_rect<ZB_PITCH_BYTES, true>(zb, key, row_bcast<J>(mine.geom[1]), row_bcast<J>(mine.shape[1]),
                                      row_bcast<J>(mine.split), ly);
}

__device__ __forceinline__ void stage_rect(unsigned &geom, unsigned &shape, unsigned base, unsigned pitch_bytes,
                                           unsigned xa, unsigned xb, unsigned row, unsigned h)
{
    const unsigned lg = raster_lg(h);
    geom = (base + xa * pitch_bytes + (row << 3)) | ((base + xb * pitch_bytes) << 16);
    shape = h == 0u ? 0u : h | (lg << 8) | (((16u * pitch_bytes) >> lg) << 16);
}

// The rectangles of an entry clipped to the tile at (x0, y0); all shapes are 0
// when the entry is absent or misses the tile.
__device__ __forceinline__ StagedEntry stage_entry(const uint4 &raw, int x0, int y0, bool valid)
{
    int xa = (int)(raw.z & 0xffffu) - x0, xb = (int)(raw.z >> 16) - x0;
    int ya = (int)(raw.w & 0xffffu) - y0, yb = (int)(raw.w >> 16) - y0;
    xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya;
    xb = xb > TILE_X ? TILE_X : xb; yb = yb > TILE_Y ? TILE_Y : yb;
    StagedEntry e;
    e.klo = raw.x; e.khi = raw.y;
    if (!valid || xb <= xa || yb <= ya) { xb = xa = 0; yb = ya = 0; }
    int lo, hi;
    cell_rows(ya, yb, lo, hi);
    if (hi > lo) {
        stage_rect(e.geom[0], e.shape[0], ZC_BASE_BYTES, ZC_PITCH_BYTES, xa, xb, lo >> ZB_CELL_LOG, (hi - lo) >> ZB_CELL_LOG);
        stage_rect(e.geom[1], e.shape[1], 0u, ZB_PITCH_BYTES, xa, xb, ya, (lo - ya) + (yb - hi));
        e.split = (unsigned)(lo - ya) | ((unsigned)(hi - lo) << (3 + 8));
    } else {
        e.geom[0] = e.shape[0] = 0u;
        stage_rect(e.geom[1], e.shape[1], 0u, ZB_PITCH_BYTES, xa, xb, ya, yb - ya);
        e.split = 0xffu;                       // no row reaches the split
    }
    return e;
}

// Scan of the [tile][class] counters, one tile per thread: exclusive offsets,
// grand total, and the list of non-empty tiles with their entry ranges, which
// is what the z-buffer kernel iterates over.  Every 1024-thread block first
// re-reads the counters of all tiles (a few hundred KB of L2 hits), so a single
// launch needs no inter-block ordering.  tile_cur is cleared here for the fill
// pass.
//
// The list is ordered by decreasing raster cost in half-octave buckets (the
// order inside a bucket is arbitrary): the persistent workgroups of the
// z-buffer kernel take tiles front to back, so the expensive tiles start first
// and the cheap ones even out the finish times.
struct TileDesc { unsigned begin, end, view, origin; };   // origin = x0 | y0 << 16
constexpr int ZB_BUCKETS = 64;
#ifndef MVF_ZB_TILE_FIXED_COST
#define MVF_ZB_TILE_FIXED_COST 256
#endif
constexpr unsigned ZB_TILE_FIXED_COST = MVF_ZB_TILE_FIXED_COST;    // readout + reset of a tile, in raster steps

// Entries of a tile (low 32 bits) and its cost bucket + 1 (0 = empty tile) above.
__device__ __forceinline__ u64 tile_weight(const uint4 &a, const uint4 &b)
{
    const u64 n = (u64)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    if (n == 0) return 0;
    // class k costs about 2^(NCLS - 1 - k) raster steps per entry
    u64 w = ((u64)a.x << 7) + ((u64)a.y << 6) + ((u64)a.z << 5) + ((u64)a.w << 4) +
            ((u64)b.x << 3) + ((u64)b.y << 2) + ((u64)b.z << 1) + (u64)b.w + ZB_TILE_FIXED_COST;
    const int l = 63 - __clzll((long long)w);
    int hb = 2 * l + (int)((w >> (l - 1)) & 1u);               // half octaves, >= 16
    hb = hb > ZB_BUCKETS - 1 ? ZB_BUCKETS - 1 : hb;
    return n | ((u64)(ZB_BUCKETS - hb) << 32);                  // bucket 0 = most expensive
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(
    const unsigned *__restrict__ tile_count, i64 ntiles, unsigned *__restrict__ tile_off,
    unsigned *__restrict__ tile_cur, TileDesc *__restrict__ descs, unsigned *__restrict__ n_desc,
    i64 *__restrict__ total_out, int tiles_y, int tiles_per_view)
{
    __shared__ unsigned h_all[ZB_BUCKETS], h_front[ZB_BUCKETS], h_blk[ZB_BUCKETS], h_base[ZB_BUCKETS];
    if (threadIdx.x < ZB_BUCKETS) h_all[threadIdx.x] = h_front[threadIdx.x] = h_blk[threadIdx.x] = 0u;
    __syncthreads();
    const i64 first = (i64)blockIdx.x * 1024;
    const i64 t = first + threadIdx.x;
    u64 front = 0, own = 0;
    uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
    for (i64 p = threadIdx.x; p < ntiles; p += 1024) {
        const uint4 pa = *(const uint4 *)(tile_count + p * NCLS);
        const uint4 pb = *(const uint4 *)(tile_count + p * NCLS + 4);
        const u64 w = tile_weight(pa, pb);
        if (p == t) { a = pa; b = pb; own = w; }
        if (w) {
            const unsigned bk = (unsigned)(w >> 32) - 1u;
            atomicAdd(&h_all[bk], 1u);
            if (p < first) { atomicAdd(&h_front[bk], 1u); front += w & 0xffffffffu; }
        }
    }
    u64 front_total, own_total;
    block_exclusive_scan<u64, 16>(front, front_total);
    const u64 ex = block_exclusive_scan<u64, 16>(own & 0xffffffffu, own_total) + front_total;
    // bucket bases (block_exclusive_scan ends with a barrier, h_all is complete)
    if (threadIdx.x < 64) {
        const unsigned c = h_all[threadIdx.x];
        unsigned inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(inc, o, 64);
            if ((int)threadIdx.x >= o) inc += up;
        }
        h_base[threadIdx.x] = inc - c;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 63) *n_desc = inc;
    }
    __syncthreads();
    if (t < ntiles) {
        const unsigned c[NCLS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned run = (unsigned)ex;
        const unsigned begin = run;
        unsigned o[NCLS];
#pragma unroll
        for (int k = 0; k < NCLS; ++k) { o[k] = run; run += c[k]; }
        *(uint4 *)(tile_off + t * NCLS) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4 *)(tile_off + t * NCLS + 4) = make_uint4(o[4], o[5], o[6], o[7]);
        *(uint4 *)(tile_cur + t * NCLS) = make_uint4(0u, 0u, 0u, 0u);
        *(uint4 *)(tile_cur + t * NCLS + 4) = make_uint4(0u, 0u, 0u, 0u);
        if (own) {
            const unsigned bk = (unsigned)(own >> 32) - 1u;
            const unsigned slot = h_base[bk] + h_front[bk] + atomicAdd(&h_blk[bk], 1u);
            const unsigned v = (unsigned)t / (unsigned)tiles_per_view, lt = (unsigned)t - v * tiles_per_view;
            const unsigned tx = lt / (unsigned)tiles_y, ty = lt - tx * tiles_y;
            descs[slot] = TileDesc{begin, run, v, (tx << TILE_X_LOG) | (ty << (TILE_Y_LOG + 16))};
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const u64 tot = front_total + own_total;
        tile_off[ntiles * NCLS] = (unsigned)tot;
        if (total_out) *total_out = (i64)tot;
    }
}

// Persistent workgroups: workgroup b rasterises the non-empty tiles b, b + G,
// b + 2G ... (G = grid size) and fetches the next tile's descriptor and first
// entries while the current tile is still being rasterised, so the per-tile
// latency chain (descriptor -> entries) is off the critical path.
// MODE 0: mark seen[winner][v] = 1 (seen_stride bytes per point).
// MODE 1: store winner + 1 of every won pixel to idmap[v][x][y] (the caller
// pre-zeroes idmap).  MODE 2: both, and the empty pixels of a non-empty tile
// are stored as 0, so that the tiles a splat touches are complete without any
// clearing of idmap.
constexpr int ZB_THREADS = TILE_X * TILE_Y / 8;  // one 8-pixel column piece per thread
#ifndef MVF_ZB_WG_PER_CU
#define MVF_ZB_WG_PER_CU (MVF_TILE_Y_LOG == 6 ? 3 : 6)
#endif
constexpr int ZB_WG_PER_CU = MVF_ZB_WG_PER_CU;  // 50 KB of LDS each
constexpr int ZB_QUARTERS = ZB_THREADS / 16;

#ifdef MVF_ZB_DEBUG
__device__ int g_zb_debug;     // bit 0: no raster, bit 1: no winner stores, bit 2: no readout
#define ZBD(bit) (dbg_ & (bit))
#else
#define ZBD(bit) false
#endif
#ifdef MVF_ZB_TIMING
__device__ unsigned long long g_zb_timing[8];
#define ZBT(i) do { const long long now_ = clock64(); acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define ZBT(i)
#endif
#ifdef MVF_ZB_SPAN
__device__ unsigned long long g_zb_span[2048 * 6];
__device__ unsigned g_zb_tile[16384 * 2];
__device__ unsigned g_zb_log[1024 * 16 * 3];            // per workgroup and tile: list position, end time, entries             // per list position: duration (10 ns), entries   // per workgroup: first / last wall clock, tiles done
#endif

template <int MODE>
#ifdef MVF_ZB_SPAN
__attribute__((amdgpu_num_sgpr(76)))
#endif
__global__ __launch_bounds__(ZB_THREADS) void tile_zbuffer_kernel(
    const TileEntry *__restrict__ entries, const TileDesc *__restrict__ descs,
    const unsigned *__restrict__ n_desc, unsigned *__restrict__ queue, i64 pair_capacity,
    int tiles_y, int tiles_per_view, int W, int Hc, uint8_t *__restrict__ seen, i64 seen_stride,
    unsigned *__restrict__ idmap, const int *__restrict__ view_first)
{
    __shared__ __align__(16) u64 zb[ZB_KEYS];
    __shared__ unsigned s_next;
#ifdef MVF_ZB_PRIO
    __builtin_amdgcn_s_setprio(MVF_ZB_PRIO);
#endif
    const unsigned nd = *n_desc;
    if (blockIdx.x >= nd) return;
#ifdef MVF_ZB_DEBUG
    const int dbg_ = g_zb_debug;
#endif
    // Tiles are handed out front to back in the cost-ordered list.  A workgroup
    // always knows its next two tiles: the first three are blockIdx, + G and
    // + 2G, the following ones come from one of ZB_QUEUES ticket counters
    // (counter q hands out the tiles 3G + q, 3G + q + ZB_QUEUES, ...; a
    // workgroup stays with counter blockIdx % ZB_QUEUES and stops when that one
    // runs dry).  The descriptor of the tile after the next is requested while
    // the current tile is rasterised, and so are the first entries of the next
    // tile, so neither load is waited for.
    unsigned k1 = blockIdx.x + gridDim.x, k2 = k1 + gridDim.x;
    const unsigned q = blockIdx.x % ZB_QUEUES;
    queue += q * ZB_QUEUE_STRIDE;
    const int tid = threadIdx.x;
    const int ly = tid & 15;
    // Entry of this lane inside a group: round ly, quarter tid >> 4.  Entries are
    // ordered by decreasing cost, so odd rounds deal the quarters in reverse
    // order (the quarter that got the most expensive entry of a round gets the
    // cheapest one of the next).
    const int slot = ly * ZB_QUARTERS + ((ly & 1) ? ZB_QUARTERS - 1 - (tid >> 4) : (tid >> 4));

#ifdef MVF_ZB_TIMING
    long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
#ifdef MVF_ZB_SPAN
    const unsigned long long t_first_ = wall_clock64();
    unsigned tiles_done_ = 0;
    unsigned long long t_dry_ = 0, t_tile_ = t_first_, last_dur_ = 0, last_n_ = 0;
    unsigned kc_ = blockIdx.x;
#endif
    TileDesc d = descs[blockIdx.x];
    TileDesc d1 = d;
    if (k1 < nd) d1 = descs[k1];
    i64 begin = d.begin, end = d.end;
    if (end > pair_capacity) end = pair_capacity;
    if (begin > end) begin = end;
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
    if (begin + slot < end) raw = *(const uint4 *)(entries + begin + slot);
    for (int p = tid; p < (int)ZB_KEYS; p += ZB_THREADS) zb[p] = ZB_EMPTY;
    __syncthreads();
    ZBT(0);

    while (true) {
        const bool more = k1 < nd;
        i64 nbegin = d1.begin, nend = d1.end;          // requested one tile ago
        if (nend > pair_capacity) nend = pair_capacity;
        if (nbegin > nend) nbegin = nend;
        unsigned ticket = 0;
        TileDesc d2 = d1;
        uint4 raw_n = make_uint4(0u, 0u, 0u, 0u);

        const int v = d.view;
        // seen flags are indexed by the view's place among the views of its sample
        const int lv = view_first ? v - view_first[v] : v;
        const int x0 = d.origin & 0xffffu, y0 = d.origin >> 16;

        i64 base = begin;
        do {
            const i64 rem = end - base;
            ZBT(1);
            const StagedEntry mine = stage_entry(raw, x0, y0, slot < rem);
            ZBT(2);
            uint4 nxt = make_uint4(0u, 0u, 0u, 0u);
            if (base + ZB_THREADS < end) {
                // prefetch the next group
                if (base + ZB_THREADS + slot < end) nxt = *(const uint4 *)(entries + base + ZB_THREADS + slot);
            } else {
                // last group: request what the coming tiles need (their first
                // entries, a ticket, a descriptor); nothing of it is looked
                // at before the raster loop below is through
                if (more && nbegin + slot < nend) raw_n = *(const uint4 *)(entries + nbegin + slot);
#ifdef MVF_ZB_DEBUG
                if (ZBD(8)) ticket = 0;            // static round-robin instead of tickets (timing only)
                else
#endif
                if (more && tid == 0) ticket = atomicAdd(queue, 1u);
                if (k2 < nd) d2 = descs[k2];
            }
            const int maxj = rem >= ZB_THREADS ? 16 : (int)((rem + ZB_QUARTERS - 1) / ZB_QUARTERS);   // lanes in use per quarter
#define MVF_RASTER(J) if (J < maxj && !ZBD(1)) raster_entry<J>((char *)zb, mine, (unsigned)ly);
            MVF_RASTER(0) MVF_RASTER(1) MVF_RASTER(2) MVF_RASTER(3)
            MVF_RASTER(4) MVF_RASTER(5) MVF_RASTER(6) MVF_RASTER(7)
            MVF_RASTER(8) MVF_RASTER(9) MVF_RASTER(10) MVF_RASTER(11)
            MVF_RASTER(12) MVF_RASTER(13) MVF_RASTER(14) MVF_RASTER(15)
#undef MVF_RASTER
            raw = nxt;
            base += ZB_THREADS;
            ZBT(3);
        } while (base < end);
#ifdef MVF_ZB_TIMING
        __builtin_amdgcn_s_waitcnt(0);
        ZBT(4);
#endif
        __syncthreads();
        ZBT(5);

        // read out the tile and reset it for the next one: every thread owns 8
        // vertically adjacent pixels (one 64-byte piece of a column) and the two
        // cells above them.  A row of 16 lanes holds the same rows of 16
        // neighbouring columns, so the winner to the left of a pixel is a DPP
        // shift away.
        if (!ZBD(4)) {
            static_assert(ZB_CELL == 4, "two cells per piece");
            constexpr int RBS = TILE_Y / 8;                              // pieces per column
            const int rb = (tid >> 4) & (RBS - 1);                       // rows 8 rb .. 8 rb + 7
            const int col = ((tid >> 4) / RBS) * 16 + (tid & 15);
            u64 *px = zb + col * ZB_PITCH + rb * 8;
            u64 *cl = zb + TILE_X * ZB_PITCH + col * ZC_PITCH + rb * 2;
            u64 key[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) key[k] = px[k];
            const u64 c0 = cl[0], c1 = cl[1];
#pragma unroll
            for (int k = 0; k < 8; ++k) px[k] = ZB_EMPTY;
            cl[0] = ZB_EMPTY; cl[1] = ZB_EMPTY;
            unsigned id[8];                                   // empty pixels give 0xffffffff
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u64 c = k < 4 ? c0 : c1;
                id[k] = (unsigned)(key[k] < c ? key[k] : c);
            }
            if (MODE != 0) {
                const int gx = x0 + col, gy = y0 + rb * 8;
                if (gx < W) {
                    unsigned *im = idmap + (size_t)v * W * Hc + (size_t)gx * Hc + gy;
                    if (MODE == 2 && (Hc & 3) == 0 && gy + 8 <= Hc) {
                        // the piece is 32 contiguous, 16-byte aligned bytes of the id map
                        ((uint4 *)im)[0] = make_uint4(id[0] + 1u, id[1] + 1u, id[2] + 1u, id[3] + 1u);
                        ((uint4 *)im)[1] = make_uint4(id[4] + 1u, id[5] + 1u, id[6] + 1u, id[7] + 1u);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if ((MODE == 2 || id[k] != 0xffffffffu) && gy + k < Hc) im[k] = id[k] + 1u;
                    }
                }
            }
            if (MODE != 1) {
                // A winner is marked where neither the pixel above (inside the
                // piece) nor the pixel to the left (the first lane of a row goes
                // without) shows it: its top-left visible pixel always qualifies.
                uint8_t *sv = seen + lv;
                unsigned prev = 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // row_shr:1, lanes without a source keep "empty"
                    const unsigned left = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)id[k], 0x111, 0xf, 0xf, false);
                    if (id[k] != prev && id[k] != left && id[k] != 0xffffffffu && !ZBD(2))
                        sv[(size_t)id[k] * seen_stride] = 1;
                    prev = id[k];
                }
            }
        }
        ZBT(6);
#ifdef MVF_ZB_SPAN
        ++tiles_done_;
        if (k2 >= nd && t_dry_ == 0) t_dry_ = wall_clock64() | ((unsigned long long)tiles_done_ << 56);
        { const unsigned long long now_ = wall_clock64(); last_dur_ = now_ - t_tile_; t_tile_ = now_; last_n_ = (unsigned long long)(end - begin);
          if (tid == 0 && blockIdx.x < 1024 && tiles_done_ <= 16) { unsigned *lg_ = g_zb_log + (blockIdx.x * 16 + tiles_done_ - 1) * 3; lg_[0] = kc_; lg_[1] = (unsigned)(now_ - t_first_); lg_[2] = (unsigned)last_n_; }
          if (tid == 0 && kc_ < 16384) { g_zb_tile[kc_ * 2] = (unsigned)last_dur_; g_zb_tile[kc_ * 2 + 1] = (unsigned)last_n_; } }
#endif
        if (!more) break;
#ifdef MVF_ZB_DEBUG
        if (ZBD(8)) { if (tid == 0) s_next = k2 + gridDim.x; } else
#endif
        if (tid == 0) s_next = 3u * gridDim.x + q + ticket * ZB_QUEUES;
#ifdef MVF_ZB_SPAN
        kc_ = k1;
#endif
        k1 = k2; d = d1; d1 = d2; begin = nbegin; end = nend; raw = raw_n;
        __syncthreads();
        k2 = s_next;
        ZBT(7);
    }
#ifdef MVF_ZB_TIMING
    if ((tid & 63) == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_zb_timing[i], (unsigned long long)acc_[i]);
#endif
#ifdef MVF_ZB_SPAN
    if (tid == 0 && blockIdx.x < 2048) {
        g_zb_span[blockIdx.x * 6] = t_first_;
        g_zb_span[blockIdx.x * 6 + 1] = wall_clock64();
        g_zb_span[blockIdx.x * 6 + 2] = tiles_done_;
        g_zb_span[blockIdx.x * 6 + 3] = t_dry_;
        g_zb_span[blockIdx.x * 6 + 4] = last_dur_;
        g_zb_span[blockIdx.x * 6 + 5] = last_n_;
    }
#endif
}

#ifdef MVF_ZB_DEBUG
extern "C" int mvf_debug_zb_switch(int mask)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_zb_debug), &mask, sizeof(mask)) != hipSuccess;
}
#endif

#ifdef MVF_ZB_TIMING
extern "C" int mvf_debug_zb_timing(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_zb_timing), sizeof(g_zb_timing)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_zb_timing), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifdef MVF_ZB_SPAN
extern "C" int mvf_debug_zb_tiles(unsigned *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_zb_tile), sizeof(unsigned) * 2 * n) != hipSuccess;
}
extern "C" int mvf_debug_zb_log(unsigned *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_zb_log), sizeof(unsigned) * 1024 * 16 * 3) != hipSuccess;
}
extern "C" int mvf_debug_zb_span(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_zb_span), sizeof(unsigned long long) * 6 * n) != hipSuccess;
}
#endif

// Counts what tile_zbuffer_kernel issues for the bins of a call, without
// touching any z-buffer: out[0] += LDS atomic wave-instructions (a loop step of
// a wave counts once, whatever the number of active lanes), out[1] += lane-level
// atomics (pixels and cells written), out[2] += (splat, tile) entries.  Mirrors
// the dealing of entries to quarters and the rectangle sweep of raster_rect.
template <unsigned PITCH_BYTES, bool SPLIT>
__device__ __forceinline__ unsigned rect_trips(unsigned geom, unsigned shape, unsigned split, unsigned ly)
{
    const unsigned h = shape & 0xffu;
    const unsigned lg = (shape >> 8) & 0xffu;
    const unsigned xo = ly >> lg, yo = ly - (xo << lg);
    if (yo >= h) return 0u;
    const unsigned step = shape >> 16, end = geom >> 16;
    unsigned a = (geom & 0xffffu) + xo * PITCH_BYTES + (yo << 3);
    if (SPLIT) a += yo >= (split & 0xffu) ? split >> 8 : 0u;
    return a < end ? (end - a + step - 1u) / step : 0u;
}

__device__ __forceinline__ unsigned wave_max(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(ZB_THREADS) void zbuffer_count_kernel(
    const TileEntry *__restrict__ entries, const TileDesc *__restrict__ descs,
    const unsigned *__restrict__ n_desc, i64 pair_capacity, u64 *__restrict__ out, unsigned max_desc)
{
    // (never past the descriptor list, whatever a workspace of another pass holds there)
    const unsigned nd = *n_desc < max_desc ? *n_desc : max_desc;
    const int tid = threadIdx.x;
    const unsigned ly = tid & 15, lane = tid & 63;
    const int slot = ly * ZB_QUARTERS + ((ly & 1) ? ZB_QUARTERS - 1 - (tid >> 4) : (tid >> 4));
    u64 instr = 0, lanes = 0, ents = 0;
    for (unsigned k = blockIdx.x; k < nd; k += gridDim.x) {
        const TileDesc d = descs[k];
        i64 begin = d.begin, end = d.end;
        if (end > pair_capacity) end = pair_capacity;
        if (begin > end) begin = end;
        const int x0 = d.origin & 0xffffu, y0 = d.origin >> 16;
        for (i64 base = begin; base < end; base += ZB_THREADS) {
            const i64 rem = end - base;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (slot < rem) raw = *(const uint4 *)(entries + base + slot);
            const StagedEntry mine = stage_entry(raw, x0, y0, slot < rem);
            ents += wave_sum(slot < rem ? 1u : 0u);
            const int maxj = rem >= ZB_THREADS ? 16 : (int)((rem + ZB_QUARTERS - 1) / ZB_QUARTERS);
            for (int j = 0; j < maxj; ++j) {
                const int from = (int)(lane & 48u) + j;              // lane j of the quarter's row
                const unsigned t0 = rect_trips<ZC_PITCH_BYTES, false>(
                    __shfl(mine.geom[0], from, 64), __shfl(mine.shape[0], from, 64), 0u, ly);
                const unsigned t1 = rect_trips<ZB_PITCH_BYTES, true>(
                    __shfl(mine.geom[1], from, 64), __shfl(mine.shape[1], from, 64),
                    __shfl(mine.split, from, 64), ly);
                instr += wave_max(t0) + wave_max(t1);
                lanes += wave_sum(t0 + t1);
            }
        }
    }
    if (lane == 0) { atomicAdd(&out[0], instr); atomicAdd(&out[1], lanes); atomicAdd(&out[2], ents); }
}

// out[3] += in-view (point, view) pairs of the call: the records the count pass staged.
__global__ __launch_bounds__(256) void staged_total_kernel(
    const unsigned *__restrict__ staged_count, i64 blocks, u64 *__restrict__ out)
{
    u64 s = 0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < blocks; k += (i64)gridDim.x * 256) s += staged_count[k];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&out[3], s);
}

// ------------------------------------------------------------------------- //
// exact mode for the single-view API: stamp splat centres of seen points      //
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void centre_stamp_kernel(
    const double *__restrict__ x, const double *__restrict__ y, i64 m,
    const uint8_t *__restrict__ seen, int crop_top, int Hc, unsigned *__restrict__ idmap)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !seen[i]) return;
    const int cx = (int)x[i], cy = (int)y[i] - crop_top;
    atomicMax(&idmap[(size_t)cx * Hc + cy], (unsigned)i + 1u);   // larger index overwrites
}

struct IdmapPred {
    const unsigned *idmap;
    __device__ bool operator()(i64 p) const { return idmap[p] != 0u; }
};
struct IdmapEmit {
    const unsigned *idmap; int Hc, crop_top; i64 *idx, *x, *y;
    __device__ void operator()(i64 p, i64 pos) const {
        const i64 px = p / Hc;
        idx[pos] = (i64)idmap[p] - 1; x[pos] = px; y[pos] = p - px * Hc + crop_top;
    }
};

// ------------------------------------------------------------------------- //
// depth-map visibility                                                        //
// ------------------------------------------------------------------------- //
struct DepthPred {
    const double *x, *y; const float *dist, *map; int H; float thr;
    __device__ bool operator()(i64 i) const {
        const i64 xi = (i64)x[i], yi = (i64)y[i];
        const float real = map[(size_t)xi * H + yi];
        return fabsf(__fsub_rn(real, dist[i])) <= thr;
    }
};
struct DepthEmit {
    const double *x, *y; i64 *oidx; double *ox, *oy;
    __device__ void operator()(i64 i, i64 pos) const { oidx[pos] = i; ox[pos] = x[i]; oy[pos] = y[i]; }
};

// ------------------------------------------------------------------------- //
// mapping features (postprocess_features, visibility.py:1548-1600)            //
// ------------------------------------------------------------------------- //
struct FeatArgs {
    const float *lin, *pla, *sca, *nrm;  // per point, may be null
    int F, H; float lo, hi;              // r_min, r_max + 1e-4 as float32
};

__device__ __forceinline__ void write_features(
    float *__restrict__ o, const FeatArgs &A, i64 pt, float px, float py, float pz,
    const float *img_xyz, float dist, double y_proj)
{
    int c = 0;
    o[c++] = __fdiv_rn(__fsub_rn(dist, A.lo), A.hi);
    if (A.lin) o[c++] = A.lin[pt];
    if (A.pla) o[c++] = A.pla[pt];
    if (A.sca) o[c++] = A.sca[pt];
    if (A.nrm) {
        const float den = __fadd_rn(dist, 1e-4f);
        const float u0 = __fdiv_rn(__fsub_rn(px, img_xyz[0]), den);
        const float u1 = __fdiv_rn(__fsub_rn(py, img_xyz[1]), den);
        const float u2 = __fdiv_rn(__fsub_rn(pz, img_xyz[2]), den);
        const float a = __fmul_rn(u0, A.nrm[3 * pt]), b = __fmul_rn(u1, A.nrm[3 * pt + 1]),
                    cc = __fmul_rn(u2, A.nrm[3 * pt + 2]);
        o[c++] = fabsf(__fadd_rn(__fadd_rn(a, b), cc));
    }
    o[c++] = (float)(y_proj / (double)A.H);
}

struct Float3 { float v[3]; };

__global__ __launch_bounds__(256) void features_kernel(
    const float *__restrict__ xyz, const i64 *__restrict__ idx, const double *__restrict__ y_proj,
    const float *__restrict__ dist, i64 k, Float3 img_xyz, FeatArgs A, float *__restrict__ out)
{
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    const i64 pt = idx[j];
    write_features(out + (size_t)j * A.F, A, pt, xyz[3 * pt], xyz[3 * pt + 1], xyz[3 * pt + 2],
                   img_xyz.v, dist[j], y_proj[j]);
}

// ------------------------------------------------------------------------- //
// fused MapImages (exact): centre conflicts + CSR assembly                    //
// ------------------------------------------------------------------------- //
// After the tile pass seen[i][v] (one byte per view, rows of seen_stride bytes,
// a multiple of 16) tells which points own a z-buffer pixel.
// Seen points that share a centre pixel of a view: the one with the highest
// local index keeps the pixel -- the reference's ascending overwrite
// (visibility.py:1183-1187).  The seen (view, pixel) pairs are few against the
// V * W * Hc pixels, so they go into an open-addressing hash table (linear
// probing, sized by the caller to a few times the number of mappings, cleared
// by the count pass).  A slot is one 64-bit word, pixel id + 1 above index + 1:
// a point claims an empty slot with one compare-and-swap; only points that
// really share a pixel need a second atomic (a maximum, which keeps the pixel
// id and raises the index).
struct CentreTable { u64 *slots; unsigned mask; int shift; };       // mask + 1 = 2^(64 - shift) slots
constexpr int CENTRE_MAX_PROBES = 512;

__device__ __forceinline__ unsigned centre_pixel_id(int v, int cx, int cy, int W, int Hc)
{
    return ((unsigned)v * (unsigned)W + (unsigned)cx) * (unsigned)Hc + (unsigned)cy + 1u;   // host checks V W Hc < 2^32 - 1
}
__device__ __forceinline__ unsigned centre_hash(const CentreTable &T, unsigned id)
{
    return (unsigned)(((u64)id * 0x9E3779B97F4A7C15ull) >> T.shift);
}
// Registers index i at pixel `id`; false when the table is too full.
__device__ __forceinline__ bool centre_insert(const CentreTable &T, unsigned id, unsigned i)
{
    const u64 mine = ((u64)id << 32) | (u64)(i + 1u);
    unsigned s = centre_hash(T, id);
    for (int probe = 0; probe < CENTRE_MAX_PROBES; ++probe, s = (s + 1u) & T.mask) {
        const u64 old = atomicCAS(&T.slots[s], 0ull, mine);
        if (old == 0ull) return true;
        if ((unsigned)(old >> 32) == id) { atomicMax(&T.slots[s], mine); return true; }
    }
    return false;
}
// Highest index + 1 registered at pixel `id` (0 = none).
__device__ __forceinline__ unsigned centre_lookup(const CentreTable &T, unsigned id)
{
    unsigned s = centre_hash(T, id);
    for (int probe = 0; probe < CENTRE_MAX_PROBES; ++probe, s = (s + 1u) & T.mask) {
        const u64 k = T.slots[s];
        if ((unsigned)(k >> 32) == id) return (unsigned)k;
        if (k == 0ull) return 0u;
    }
    return 0u;
}

// Bit mask of the views [16 * chunk, 16 * chunk + 16) that see point i.
__device__ __forceinline__ unsigned seen_chunk(const uint8_t *seen, i64 seen_stride, i64 i, int chunk)
{
    const uint4 f = *(const uint4 *)(seen + (size_t)i * seen_stride + 16 * chunk);
    const unsigned w[4] = {f.x, f.y, f.z, f.w};
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // flags are 0 or 1: gather the low bit of each byte
        const unsigned t = w[k] & 0x01010101u;
        m |= (((t * 0x01020408u) >> 24) & 0xfu) << (4 * k);
    }
    return m;
}

// Copies the views of a call into LDS (16 bytes per thread step) and returns the
// source with its view table there: the per-lane, per-view camera constants of
// the kernels below are then LDS reads instead of divergent global loads.
template <class Src>
__device__ __forceinline__ Src views_to_lds(Src src, int V, mvf_view *lds_views)
{
    const uint4 *g = (const uint4 *)src.views;
    uint4 *l = (uint4 *)lds_views;
    for (int k = threadIdx.x; k < V * (int)(sizeof(mvf_view) / 16); k += blockDim.x) l[k] = g[k];
    __syncthreads();
    src.views = lds_views;
    return src;
}

template <class Src>
__global__ __launch_bounds__(256) void mi_centre_kernel(
    Src src, mvf_vis_params P, const uint8_t *__restrict__ seen, i64 seen_stride, int V, int Hc,
    CentreTable table, i64 *__restrict__ overflow)
{
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    src = views_to_lds(src, V, (mvf_view *)dyn_lds);
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= src.n) return;
    const int2 win = src.views_of(i, V);
    for (int chunk = 0; chunk * 16 < win.y; ++chunk) {
        unsigned m = seen_chunk(seen, seen_stride, i, chunk);
        while (m) {
            const int v = win.x + 16 * chunk + __ffs(m) - 1;
            m &= m - 1;
            double x, y, fx, fy; float d;
            if (!src.get(i, v, P, x, y, d, fx, fy)) continue;
            const int cx = (int)x, cy = (int)y - P.crop_top;
            if (!centre_insert(table, centre_pixel_id(v, cx, cy, P.W, Hc), (unsigned)i))
                atomicOr((unsigned long long *)overflow, 2ull);
        }
    }
}

struct PixelXform { int up, down, crop_w, crop_h; };

// The kernels of the fused paths keep the views of a call in LDS (128 bytes each).
constexpr int MI_MAX_VIEWS = 480;

// Which (point, view) pairs the visibility model keeps.  `candidates`: bit mask
// of the views [32 wd, 32 wd + 32) worth testing for point position i;
// `keeps`: the model's verdict for a pair that passed the projection.
struct SplatKeep {
    // exact splat visibility: candidates are the winners of the z-buffer, kept is
    // who holds its centre pixel (visibility.py:1168-1187)
    const uint8_t *seen; i64 seen_stride; CentreTable table; int Hc;
    __device__ unsigned candidates(i64 i, int wd, int V) const {
        unsigned sm = seen_chunk(seen, seen_stride, i, 2 * wd);
        if (32 * wd + 16 < V) sm |= seen_chunk(seen, seen_stride, i, 2 * wd + 1) << 16;
        return sm;
    }
    __device__ bool keeps(const mvf_vis_params &P, i64 i, int v, int cx, int cy, float) const {
        return centre_lookup(table, centre_pixel_id(v, cx, cy - P.crop_top, P.W, Hc)) == (unsigned)i + 1u;
    }
};
struct DepthKeep {
    // visibility_from_depth_map (visibility.py:1361-1388): the map's depth at the
    // pixel is within depth_threshold of the point's distance; maps [V][W][H]
    const float *maps;
    __device__ unsigned candidates(i64, int wd, int V) const {
        const int left = V - 32 * wd;
        return left >= 32 ? 0xffffffffu : (1u << left) - 1u;
    }
    __device__ bool keeps(const mvf_vis_params &P, i64, int v, int cx, int cy, float dist) const {
        const float real = maps[((size_t)v * P.W + cx) * P.H + cy];
        return fabsf(__fsub_rn(real, dist)) <= P.depth_threshold;
    }
};

// Shared by the count and the fill pass: does the point position i keep its
// mapping in view v, and at which (converted) pixel?  0 = not mapped, 1 =
// mapped by the visibility model but outside the crop (the view still counts
// as seen, image.py:289-290,350), 2 = kept.
template <class Keep, class Src>
__device__ __forceinline__ int mi_survives(
    const Src &src, const mvf_vis_params &P, const Keep &keep, const PixelXform &X, i64 i, int v,
    int &px, int &py, float &dist, double &y_proj)
{
    double x, y, fx, fy;
    if (!src.get(i, v, P, x, y, dist, fx, fy)) return 0;
    const int cx = (int)x, cy = (int)y;
    if (!keep.keeps(P, i, v, cx, cy, dist)) return 0;
    // image.py:307-319 (non-negative operands: floor division == truncation)
    px = cx / X.up - src.views[v].crop_off_x;
    py = cy / X.up - src.views[v].crop_off_y;
    if (px < 0 || py < 0 || px >= X.crop_w || py >= X.crop_h) return 1;
    px /= X.down; py /= X.down;
    y_proj = y;
    return 2;
}

// Pixel, distance and y_proj of a kept (point, view) mapping (no visibility
// re-check: the caller holds the kept-view mask).
template <class Src>
__device__ __forceinline__ void mi_pixel(
    const Src &src, const mvf_vis_params &P, const PixelXform &X, i64 i, int v,
    int &px, int &py, float &dist, double &y_proj)
{
    double x, y, fx, fy;
    src.get(i, v, P, x, y, dist, fx, fy);
    px = ((int)x / X.up - src.views[v].crop_off_x) / X.down;
    py = ((int)y / X.up - src.views[v].crop_off_y) / X.down;
    y_proj = y;
}

// Pass 1 of the CSR assembly: one thread per output point id.  Tests the
// candidate views of its point (projection, visibility model, crop), stores
// the kept-view bit masks masks[word][q] and the block's mapping count.
template <class Keep, class Src>
__global__ __launch_bounds__(256) void mi_point_mask_kernel(
    Src src, mvf_vis_params P, Keep keep, PixelXform X, int V, int Vw,
    const i64 *__restrict__ inv_ids, i64 num_points, unsigned *__restrict__ masks,
    unsigned *__restrict__ block_sums, int *__restrict__ view_seen)
{
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    src = views_to_lds(src, V, (mvf_view *)dyn_lds);
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 i = q < num_points ? inv_ids[q] : -1;
    const int2 win = i >= 0 ? src.views_of(i, V) : make_int2(0, 0);
    unsigned c = 0;
    const int words = (Vw + 31) >> 5;            // mask bits: a view's index inside the point's window
    for (int wd = 0; wd < words; ++wd) {
        unsigned sm = i >= 0 ? keep.candidates(i, wd, Vw) : 0u;
        unsigned kept = 0;
        while (sm) {
            const int bit = __ffs(sm) - 1;
            sm &= sm - 1;
            const int v = win.x + (wd << 5) + bit;
            int px, py; float d; double yp;
            const int s = mi_survives(src, P, keep, X, i, v, px, py, d, yp);
            if (s) view_seen[v] = 1;
            if (s == 2) kept |= 1u << bit;
        }
        if (q < num_points) masks[(size_t)wd * num_points + q] = kept;
        c += __popc(kept);
    }
    unsigned total;
    block_exclusive_scan<unsigned>(c, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Pass 2: pointers + images / pixels / features of every kept mapping, views
// ascending inside a point.  Every block adds up the counts of the blocks in
// front of it itself, and ranks the seen views among the seen views (lexunique
// + bucketize, image.py:392-394) in LDS; the last block also stores the totals.
template <class Src>
__global__ __launch_bounds__(256) void mi_fill_kernel(
    Src src, mvf_vis_params P, PixelXform X, int V, int Vw, const i64 *__restrict__ inv_ids,
    i64 num_points, const unsigned *__restrict__ masks, const unsigned *__restrict__ block_sums,
    const int *__restrict__ view_seen, FeatArgs A, i64 out_capacity,
    i64 *__restrict__ pointers, i64 *__restrict__ images, int16_t *__restrict__ pixels,
    float *__restrict__ features, i64 *__restrict__ counters)
{
    extern __shared__ __align__(16) unsigned char dyn_lds[];   // V views, then V + 1 ranks
    src = views_to_lds(src, V, (mvf_view *)dyn_lds);
    int *view_rank = (int *)(dyn_lds + (size_t)V * sizeof(mvf_view));
    for (int v = threadIdx.x; v <= V; v += 256) {
        int r = 0;
        for (int u = 0; u < v; ++u) r += view_seen[u] ? 1 : 0;
        view_rank[v] = r;
    }
    i64 front = 0;
    for (unsigned k = threadIdx.x; k < blockIdx.x; k += 256) front += block_sums[k];
    i64 front_total;
    block_exclusive_scan<i64>(front, front_total);       // also orders view_rank for the block

    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const int words = (Vw + 31) >> 5;
    unsigned c = 0;
    if (q < num_points)
        for (int wd = 0; wd < words; ++wd) c += __popc(masks[(size_t)wd * num_points + q]);
    unsigned total;
    i64 k = front_total + block_exclusive_scan<unsigned>(c, total);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        pointers[num_points] = front_total + total;
        counters[0] = front_total + total;
        counters[3] = view_rank[V];
    }
    if (q >= num_points) return;
    pointers[q] = k;
    if (c == 0) return;
    const i64 i = inv_ids[q];
    const int first_view = src.views_of(i, V).x;
    const float px3 = src.xyz[3 * i], py3 = src.xyz[3 * i + 1], pz3 = src.xyz[3 * i + 2];
    for (int wd = 0; wd < words; ++wd) {
        unsigned m = masks[(size_t)wd * num_points + q];
        while (m) {
            const int v = first_view + (wd << 5) + __ffs(m) - 1;
            m &= m - 1;
            if (k < out_capacity) {
                int px, py; float d; double yp;
                mi_pixel(src, P, X, i, v, px, py, d, yp);
                images[k] = view_rank[v];
                pixels[2 * k] = (int16_t)px; pixels[2 * k + 1] = (int16_t)py;
                write_features(features + (size_t)k * A.F, A, i, px3, py3, pz3,
                               src.views[v].img_xyz, d, yp);
            } else {
                atomicOr((unsigned long long *)&counters[2], 1ull);
            }
            ++k;
        }
    }
}

// ------------------------------------------------------------------------- //
// fused MapImages (plain splat visibility, exact = False): every pixel a      //
// point wins is a mapping (visibility.py:1190-1195)                           //
// ------------------------------------------------------------------------- //
// After the tile pass idmap[v][x][y] holds winner + 1 of every pixel of the
// tiles that received a splat (the other tiles hold stale words and are never
// read: the marking pass walks the list of non-empty tiles), and seen[i][v] the
// points that won a pixel.  The visible part of a splat is often a sliver of its
// box (C2: 32 M pixels won out of 170 M box pixels), and boxes range from one
// pixel to tens of thousands, so no pass walks boxes pixel by pixel:
//  1. ne_box_kernel: every seen (point, view) pair gets room for one verdict bit
//     per pixel of its splat box, as one byte ("entry") per column and group of
//     8 image rows (rows 8g .. 8g + 7), columns first: ascending (x, y) is
//     ascending (entry, bit).  Eight entries make a 64-bit word; the pair's words
//     are also cut into chunks of NE_CHUNK words, the work items of step 5.
//  2. ne_mark_kernel: a thread owns 8 vertically adjacent pixels (one row group
//     of a column), hence the entry of every winner it meets: plain byte stores.
//  3. a prefix sum of the bit counts of the words (bits_before[w] = bits of the
//     words in front of word w): the pixels in front of an entry are bits_before
//     of its word plus the bits below it in the word.  The pixels of a pair are
//     the difference over its entries.
//  4. ne_count_kernel, ne_point_kernel, ne_fill_kernel: CSR offsets.
//  5. ne_emit_kernel: a lane takes one word of a chunk, rebuilds the pixels of
//     its set bits from the pair's box, and the pixels of a chunk go out as one
//     contiguous piece of the output.  No sort, no second look at the id map.
// Pair records are laid out [view][point position].
constexpr unsigned NE_NO_MASKS = 0xffffffffu;      // as first entry: the pair's entries did not fit the buffer
#ifndef MVF_NE_CHUNK
#define MVF_NE_CHUNK 16
#endif
constexpr int NE_CHUNK = MVF_NE_CHUNK;             // words per chunk: that many neighbouring lanes take one

// x: first entry; y: first column | first row group << 16; z, w: low 32 and (in
// the low half of w) high 16 bits of the pair's first pixel in the output minus
// the pixels in front of its first entry (ne_fill_kernel); upper half of w: row
// groups per column.
typedef uint4 NePair;
__device__ __forceinline__ unsigned ne_groups(unsigned ya, unsigned yb) { return ((yb - 1u) >> 3) - (ya >> 3) + 1u; }
__device__ __forceinline__ unsigned ne_items(unsigned xa, unsigned xb, unsigned ya, unsigned yb)
{
    return xb > xa && yb > ya ? (xb - xa) * ne_groups(ya, yb) : 0u;
}
__device__ __forceinline__ unsigned ne_entry(const NePair &p, unsigned x, unsigned y)
{
    return p.x + (x - (p.y & 0xffffu)) * (p.w >> 16) + ((y >> 3) - (p.y >> 16));
}
__device__ __forceinline__ i64 ne_first_pixel(const NePair &p)
{
    return (i64)(((u64)(i64)(int16_t)(p.w & 0xffffu) << 32) | p.z);
}
// One block per NE_BOX_GROUP x 256 points and view (the records the binning count
// pass staged, NE_BOX_GROUP of its blocks).  A pair's entries start at a word (its
// count is rounded up to 8: no word holds entries of two pairs).  Entries and
// chunks are handed out with one atomic each per block (atomics on one address
// take their turns at the memory side, hence the grouping); cursor[0] ends as
// the number of entries the call needs, cursor[1] as the number of chunks.
// A chunk: x = pair, y = first word, z = words (1 .. NE_CHUNK), w = view.
constexpr int NE_BOX_GROUP = 8;
__global__ __launch_bounds__(256) void ne_box_kernel(
    const TileEntry *__restrict__ staged, const unsigned *__restrict__ staged_count, unsigned staged_blocks,
    const uint8_t *__restrict__ seen, i64 seen_stride, i64 n, NePair *__restrict__ pairs,
    uint4 *__restrict__ chunks, i64 chunk_capacity, i64 mask_capacity, u64 *__restrict__ cursor,
    i64 *__restrict__ counters)
{
    __shared__ u64 s_base[2];
    const int tid = threadIdx.x, v = blockIdx.y;
    TileEntry e[NE_BOX_GROUP];
    unsigned items[NE_BOX_GROUP];
    bool have[NE_BOX_GROUP];
#pragma unroll
    for (int g = 0; g < NE_BOX_GROUP; ++g) {
        const unsigned sb = blockIdx.x * NE_BOX_GROUP + g;
        have[g] = false;
        e[g].key = 0; e[g].xa = e[g].xb = e[g].ya = e[g].yb = 0;
        if (sb < staged_blocks) {
            const size_t blk = (size_t)v * staged_blocks + sb;
            if ((unsigned)tid < staged_count[blk]) { e[g] = staged[blk * 256 + tid]; have[g] = true; }
        }
    }
    u64 mine = 0;
#pragma unroll
    for (int g = 0; g < NE_BOX_GROUP; ++g) {
        have[g] = have[g] && seen[(size_t)(unsigned)e[g].key * seen_stride + v] != 0;
        items[g] = have[g] ? (ne_items(e[g].xa, e[g].xb, e[g].ya, e[g].yb) + 7u) & ~7u : 0u;
        mine += items[g];
    }
    u64 all;
    const u64 before = block_exclusive_scan<u64>(mine, all);
    if (all == 0) return;
    if (tid == 0) s_base[0] = atomicAdd((unsigned long long *)cursor, (unsigned long long)all);
    __syncthreads();
    // the pairs' places, and the chunks they make
    u64 mo = s_base[0] + before;
    unsigned first[NE_BOX_GROUP], my_chunks = 0;
    bool over = false;
#pragma unroll
    for (int g = 0; g < NE_BOX_GROUP; ++g) {
        first[g] = NE_NO_MASKS;
        if (items[g]) {
            if (mo + items[g] <= (u64)mask_capacity) {
                first[g] = (unsigned)mo;
                my_chunks += ((items[g] >> 3) + NE_CHUNK - 1u) / NE_CHUNK;
            } else {
                over = true;
            }
        }
        mo += items[g];
    }
    unsigned all_chunks;
    const unsigned chunks_before = block_exclusive_scan<unsigned>(my_chunks, all_chunks);
    if (tid == 0 && all_chunks) s_base[1] = atomicAdd((unsigned long long *)(cursor + 1), (unsigned long long)all_chunks);
    const bool any_over = __syncthreads_or(over);
    u64 at = s_base[1] + chunks_before;
#pragma unroll
    for (int g = 0; g < NE_BOX_GROUP; ++g) {
        if (!have[g]) continue;
        const size_t pair = (size_t)v * n + (unsigned)e[g].key;
        pairs[pair] = make_uint4(first[g], (unsigned)e[g].xa | ((unsigned)e[g].ya >> 3) << 16, 0u,
                                 (items[g] ? ne_groups(e[g].ya, e[g].yb) : 0u) << 16);
        if (first[g] == NE_NO_MASKS) continue;
        const unsigned words = items[g] >> 3;
        // (the list has room for one chunk per (splat, tile) entry the call was sized for -- a seen
        // pair has at least one -- plus the full chunks: it only runs out when those did)
        for (unsigned w = 0; w < words; w += NE_CHUNK, ++at)
            if (at < (u64)chunk_capacity)
                chunks[at] = make_uint4((unsigned)pair, (first[g] >> 3) + w,
                                        words - w < (unsigned)NE_CHUNK ? words - w : (unsigned)NE_CHUNK, (unsigned)v);
    }
    if (any_over && tid == 0) atomicOr((unsigned long long *)&counters[2], 8ull);
}

struct NeImage { const unsigned *idmap; int W, Hc, crop_top, crop_w, crop_h; i64 n; };

// A thread takes the row group rg (8 rows) of column `col` of a non-empty tile;
// a block 256 such pieces.  Adjacent lanes take adjacent row groups of a column.
constexpr int NE_MARK_BLOCKS = TILE_X * TILE_Y / 8 / 256;
__global__ __launch_bounds__(256) void ne_mark_kernel(
    const TileDesc *__restrict__ descs, const unsigned *__restrict__ n_desc, NeImage S,
    const mvf_view *__restrict__ views, const NePair *__restrict__ pairs, uint8_t *__restrict__ verdicts)
{
    constexpr int RGS = TILE_Y / 8;
    const unsigned d = blockIdx.x / NE_MARK_BLOCKS;
    if (d >= *n_desc) return;
    const TileDesc D = descs[d];
    const unsigned piece = (blockIdx.x % NE_MARK_BLOCKS) * 256u + threadIdx.x;
    const int x = (int)(D.origin & 0xffffu) + (int)(piece / RGS);
    const int y = (int)(D.origin >> 16) + (int)(piece % RGS) * 8;
    if (x >= S.W || y >= S.Hc) return;
    const int v = (int)D.view;
    const int px = x - views[v].crop_off_x, py0 = y + S.crop_top - views[v].crop_off_y;
    if (px < 0 || px >= S.crop_w) return;
    const unsigned *im = S.idmap + ((size_t)v * S.W + x) * S.Hc + y;
    unsigned val[8];
    if ((S.Hc & 3) == 0 && y + 8 <= S.Hc) {
        const uint4 a = ((const uint4 *)im)[0], b = ((const uint4 *)im)[1];
        val[0] = a.x; val[1] = a.y; val[2] = a.z; val[3] = a.w;
        val[4] = b.x; val[5] = b.y; val[6] = b.z; val[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) val[k] = y + k < S.Hc ? im[k] : 0u;
    }
    // pixels outside the crop count as empty
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (py0 + k < 0 || py0 + k >= S.crop_h) val[k] = 0u;
    // one entry per distinct winner: the first pixel of every winner leads
    NePair pr[8];
    unsigned bits[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bool lead = val[k] != 0u;
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < k && val[j] == val[k]) lead = false;
            if (val[j] == val[k]) m |= 1u << j;
        }
        bits[k] = lead ? m : 0u;
        pr[k] = make_uint4(NE_NO_MASKS, 0u, 0u, 0u);
        if (lead) pr[k] = pairs[(size_t)v * S.n + (val[k] - 1u)];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (bits[k] != 0u && pr[k].x != NE_NO_MASKS)
            verdicts[ne_entry(pr[k], (unsigned)x, (unsigned)y)] = (uint8_t)bits[k];
}

// bits_before[w] <- set bits of the 64-bit words [0, w), for w up to the number
// of words in use (inclusive): block sums, then (after a scan of the block sums)
// the words' own places.  Eight words per thread.
constexpr int NE_SCAN_WORDS = 256 * 8;
__device__ __forceinline__ i64 ne_words_used(const u64 *cursor, i64 mask_capacity)
{
    const u64 c = *cursor;
    return ((c < (u64)mask_capacity ? (i64)c : mask_capacity) + 7) >> 3;
}

__device__ __forceinline__ unsigned ne_load_words(const u64 *verdicts, i64 base, i64 n, unsigned (&c)[8])
{
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = 0u;
    if (base < n) {
        const uint4 *q = (const uint4 *)(verdicts + base);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 m = q[k];
            c[2 * k] = __popc(m.x) + __popc(m.y);
            c[2 * k + 1] = __popc(m.z) + __popc(m.w);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += c[k];
    }
    return s;
}

__global__ __launch_bounds__(256) void ne_bits_reduce_kernel(
    const u64 *__restrict__ verdicts, const u64 *__restrict__ cursor, i64 mask_capacity,
    unsigned *__restrict__ block_sums)
{
    const i64 n = ne_words_used(cursor, mask_capacity);
    const i64 base = ((i64)blockIdx.x * 256 + threadIdx.x) * 8;
    unsigned c[8];
    const unsigned s = (i64)blockIdx.x * NE_SCAN_WORDS < n ? ne_load_words(verdicts, base, n, c) : 0u;
    unsigned total;
    block_exclusive_scan<unsigned>(s, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void ne_bits_apply_kernel(
    const u64 *__restrict__ verdicts, const u64 *__restrict__ cursor, i64 mask_capacity,
    const unsigned *__restrict__ block_sums, unsigned *__restrict__ bits_before)
{
    const i64 n = ne_words_used(cursor, mask_capacity);
    if ((i64)blockIdx.x * NE_SCAN_WORDS > n) return;
    const i64 base = ((i64)blockIdx.x * 256 + threadIdx.x) * 8;
    unsigned c[8];
    const unsigned s = ne_load_words(verdicts, base, n, c);
    unsigned total;
    unsigned run = block_exclusive_scan<unsigned>(s, total) + block_sums[blockIdx.x];
    if (base > n) return;
    unsigned o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[k] = run; run += c[k]; }
    if (base + 8 <= n) {
        *(uint4 *)(bits_before + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4 *)(bits_before + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k <= n) bits_before[base + k] = o[k];
    }
}

// cnt[v][i] <- pixels of every seen pair inside the crop (same blocks as the binning passes).
__global__ __launch_bounds__(256) void ne_count_kernel(
    const TileEntry *__restrict__ staged, const unsigned *__restrict__ staged_count,
    const uint8_t *__restrict__ seen, i64 seen_stride, i64 n, const NePair *__restrict__ pairs,
    const unsigned *__restrict__ bits_before, unsigned *__restrict__ cnt)
{
    const int tid = threadIdx.x, v = blockIdx.y;
    const size_t blk = (size_t)v * gridDim.x + blockIdx.x;
    if ((unsigned)tid >= staged_count[blk]) return;
    const TileEntry e = staged[blk * 256 + tid];
    const unsigned id = (unsigned)e.key;
    if (seen[(size_t)id * seen_stride + v] == 0) return;
    const size_t pair = (size_t)v * n + id;
    const unsigned mo = pairs[pair].x;
    // (a pair's entries are whole words)
    cnt[pair] = mo == NE_NO_MASKS ? 0u
        : bits_before[(mo >> 3) + ((ne_items(e.xa, e.xb, e.ya, e.yb) + 7u) >> 3)] - bits_before[mo >> 3];
}

// pixels <- the pixels of all pairs.  A wave takes 64 / NE_CHUNK chunks (up to NE_CHUNK words
// of one pair each), a lane one word: it rebuilds the pixels of the word's set
// bits from the pair's box and lays them out, in LDS, where they stand among the
// wave's pixels; the pixels of a chunk are neighbours in the output (entry order
// is pixel order), so the wave then stores with consecutive lanes on consecutive
// pixels.  The grid is fixed; waves stride over the chunk list, fetching chunk
// and pair record one round ahead of their use.
constexpr int NE_EMIT_SLOTS = 1024;          // pixels of a wave staged at a time
__global__ __launch_bounds__(256) void ne_emit_kernel(
    const uint4 *__restrict__ chunks, const u64 *__restrict__ cursor, const NePair *__restrict__ pairs,
    const u64 *__restrict__ verdicts, const unsigned *__restrict__ bits_before,
    const mvf_view *__restrict__ views, int crop_top, unsigned *__restrict__ pixels, i64 pix_capacity,
    i64 chunk_capacity)
{
    __shared__ unsigned s_pix[4][NE_EMIT_SLOTS];
    const u64 nc = cursor[1] < (u64)chunk_capacity ? cursor[1] : (u64)chunk_capacity;
    const unsigned n_chunks = nc < 0xffffffffull ? (unsigned)nc : 0xffffffffu;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned sub = lane / NE_CHUNK, wl = lane % NE_CHUNK;
    constexpr unsigned PER_WAVE = 64 / NE_CHUNK;
    const u64 stride = (u64)gridDim.x * 4u * PER_WAVE;
    const uint4 no_chunk = make_uint4(0u, 0u, 0u, 0u);
    u64 c0 = ((u64)blockIdx.x * 4u + wave) * PER_WAVE;
    uint4 ch = c0 + sub < n_chunks ? chunks[c0 + sub] : no_chunk;
    NePair pr = wl < ch.z ? pairs[ch.x] : no_chunk;
    for (; c0 < n_chunks; c0 += stride) {
        const uint4 ch_next = c0 + stride + sub < n_chunks ? chunks[c0 + stride + sub] : no_chunk;
        u64 bits = 0;
        unsigned at = 0, col = 0, rg = 0, nrg = 1, xa = 0, g0 = 0;
        int off_x = 0, off_y = 0;
        if (wl < ch.z) {
            const unsigned w = ch.y + wl;
            nrg = pr.w >> 16;
            xa = pr.y & 0xffffu;
            g0 = pr.y >> 16;
            off_x = views[ch.w].crop_off_x;
            off_y = views[ch.w].crop_off_y - crop_top;
            bits = verdicts[w];
            at = (unsigned)(ne_first_pixel(pr) + (i64)bits_before[w]);
            const unsigned rel = (w << 3) - pr.x;           // first entry of the word, inside the pair
            col = rel / nrg;
            rg = rel - col * nrg;
        }
        const NePair pr_next = wl < ch_next.z ? pairs[ch_next.x] : no_chunk;
        ch = ch_next;
        pr = pr_next;
        // place among the wave's pixels
        const unsigned cnt = (unsigned)__popcll(bits);
        unsigned inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += o;
        }
        const unsigned total = __shfl(inc, 63, 64);
        const unsigned slot0 = inc - cnt;
        // the pixels of a chunk are consecutive in the output: first slot and first place of each
        unsigned c_slot[PER_WAVE], c_at[PER_WAVE];
#pragma unroll
        for (unsigned q = 0; q < PER_WAVE; ++q) {
            c_slot[q] = __shfl(slot0, (int)(q * NE_CHUNK), 64);
            c_at[q] = __shfl(at, (int)(q * NE_CHUNK), 64);
        }
        for (unsigned win = 0; win < total; win += NE_EMIT_SLOTS) {
            // entry by entry (a byte of the word: one column, 8 rows), all lanes in step
            unsigned slot = slot0 - win, c_ = col, r_ = rg;
#pragma unroll 1
            for (unsigned j = 0; j < 8u; ++j) {
                const unsigned byte = (unsigned)(bits >> (8u * j)) & 0xffu;
                const unsigned xy = ((unsigned)((int)(xa + c_) - off_x) & 0xffffu) | ((unsigned)((int)((g0 + r_) << 3) - off_y) << 16);
                if (__ballot(byte != 0u)) {
#pragma unroll
                    for (unsigned b = 0; b < 8u; ++b) {
                        const unsigned set = (byte >> b) & 1u;
                        if (set && slot < (unsigned)NE_EMIT_SLOTS) s_pix[wave][slot] = xy + (b << 16);
                        slot += set;
                    }
                }
                if (++r_ == nrg) { r_ = 0; ++c_; }
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned here = total - win < (unsigned)NE_EMIT_SLOTS ? total - win : (unsigned)NE_EMIT_SLOTS;
            for (unsigned t = lane; t < here; t += 64) {
                const unsigned sl = win + t;
                unsigned base_slot = 0u, base_at = c_at[0];
#pragma unroll
                for (unsigned q = 1; q < PER_WAVE; ++q)
                    if (sl >= c_slot[q]) { base_slot = c_slot[q]; base_at = c_at[q]; }
                const unsigned to = base_at + (sl - base_slot);
                if ((i64)to < pix_capacity) pixels[to] = s_pix[wave][t];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// One thread per output point id: the views that map it (won a pixel inside
// the crop), as bit masks, and the block's numbers of view rows and pixels.
__global__ __launch_bounds__(256) void ne_point_kernel(
    const uint8_t *__restrict__ seen, i64 seen_stride, const unsigned *__restrict__ cnt, i64 n, int V,
    const i64 *__restrict__ inv_ids, i64 num_points, unsigned *__restrict__ masks,
    unsigned *__restrict__ block_rows, i64 *__restrict__ block_pix, int *__restrict__ view_seen)
{
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 i = q < num_points ? inv_ids[q] : -1;
    unsigned rows = 0;
    i64 pix = 0;
    const int words = (V + 31) >> 5;
    for (int wd = 0; wd < words; ++wd) {
        unsigned sm = 0;
        if (i >= 0) {
            sm = seen_chunk(seen, seen_stride, i, 2 * wd);
            if (32 * wd + 16 < V) sm |= seen_chunk(seen, seen_stride, i, 2 * wd + 1) << 16;
        }
        unsigned kept = 0;
        while (sm) {
            const int bit = __ffs(sm) - 1;
            sm &= sm - 1;
            const int v = (wd << 5) + bit;
            view_seen[v] = 1;                       // mapped by the visibility model, cropped or not
            const unsigned c = cnt[(size_t)v * n + i];
            if (c) { kept |= 1u << bit; pix += c; }
        }
        if (q < num_points) masks[(size_t)wd * num_points + q] = kept;
        rows += __popc(kept);
    }
    unsigned rows_total;
    i64 pix_total;
    block_exclusive_scan<unsigned>(rows, rows_total);
    block_exclusive_scan<i64>(pix, pix_total);
    if (threadIdx.x == 0) { block_rows[blockIdx.x] = rows_total; block_pix[blockIdx.x] = pix_total; }
}

// f + f + ... + f (c terms, every partial sum rounded to float32, in order): what
// a sequential segment sum over c rows that all hold f returns.  Inside a binade
// the rounded increment is the same at every step once the sum has taken two
// steps there (a tie rounds to even, and from an even sum the next tie falls
// the same way), so the walk jumps through a binade with one division and
// takes real additions only around the binade changes: O(log c) steps.
__device__ __forceinline__ float repeat_add(float f, unsigned c)
{
    float s = f;
    unsigned left = c - 1u;
    if (!(f > 0.f) || (__float_as_uint(f) >> 23) == 0u || (__float_as_uint(f) >> 23) == 255u) {
        for (; left; --left) s = __fadd_rn(s, f);      // zero, negative, subnormal, not finite: plain walk
        return s;
    }
    while (left) {
        const float s1 = __fadd_rn(s, f);
        --left;
        s = s1;
        if (!left) break;
        const unsigned E = __float_as_uint(s1) >> 23;
        if (E == 255u) continue;
        const float s2 = __fadd_rn(s1, f);
        --left;
        s = s2;
        if (!left || (__float_as_uint(s2) >> 23) != E) continue;
        const float s3 = __fadd_rn(s2, f);
        --left;
        s = s3;
        if (!left || (__float_as_uint(s3) >> 23) != E) continue;
        const unsigned m2 = (__float_as_uint(s2) & 0x7fffffu) | 0x800000u;
        const unsigned m3 = (__float_as_uint(s3) & 0x7fffffu) | 0x800000u;
        const unsigned d = m3 - m2;
        if (d == 0u) break;                            // f no longer moves the sum
        const unsigned room = (0xffffffu - m3) / d;
        const unsigned k = left < room ? left : room;
        left -= k;
        s = __uint_as_float((E << 23) | ((m3 + k * d) & 0x7fffffu));
    }
    return s;
}

// Pointers, images, view pointers and view features (mean over the view's
// pixels, which all carry the features of the point, summed one by one as a
// segment mean does); the pair records receive the first pixel of the pair minus
// the bits before its first entry, for the pixel pass.
template <class Src>
__global__ __launch_bounds__(256) void ne_fill_kernel(
    Src src, mvf_vis_params P, int V, const i64 *__restrict__ inv_ids, i64 num_points,
    const unsigned *__restrict__ masks, const unsigned *__restrict__ block_rows,
    const i64 *__restrict__ block_pix, const unsigned *__restrict__ cnt, const int *__restrict__ view_seen,
    FeatArgs A, i64 out_capacity, i64 pix_capacity, i64 *__restrict__ pointers, i64 *__restrict__ images,
    i64 *__restrict__ view_ptr, float *__restrict__ features, NePair *__restrict__ pairs,
    const unsigned *__restrict__ bits_before, const u64 *__restrict__ mask_cursor, i64 *__restrict__ counters)
{
    extern __shared__ __align__(16) unsigned char dyn_lds[];   // V views, then V + 1 ranks
    src = views_to_lds(src, V, (mvf_view *)dyn_lds);
    int *view_rank = (int *)(dyn_lds + (size_t)V * sizeof(mvf_view));
    for (int v = threadIdx.x; v <= V; v += 256) {
        int r = 0;
        for (int u = 0; u < v; ++u) r += view_seen[u] ? 1 : 0;
        view_rank[v] = r;
    }
    i64 front_rows = 0, front_pix = 0;
    for (unsigned k = threadIdx.x; k < blockIdx.x; k += 256) { front_rows += block_rows[k]; front_pix += block_pix[k]; }
    i64 rows_before, pix_before;
    block_exclusive_scan<i64>(front_rows, rows_before);
    block_exclusive_scan<i64>(front_pix, pix_before);

    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 i = q < num_points ? inv_ids[q] : -1;
    const i64 n = src.count();
    const int words = (V + 31) >> 5;
    unsigned rows = 0;
    i64 pix = 0;
    if (q < num_points)
        for (int wd = 0; wd < words; ++wd) {
            unsigned m = masks[(size_t)wd * num_points + q];
            rows += __popc(m);
            while (m) {
                const int v = (wd << 5) + __ffs(m) - 1;
                m &= m - 1;
                pix += cnt[(size_t)v * n + i];
            }
        }
    unsigned rows_total;
    i64 pix_total;
    i64 k = rows_before + block_exclusive_scan<unsigned>(rows, rows_total);
    i64 p = pix_before + block_exclusive_scan<i64>(pix, pix_total);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const i64 n_rows = rows_before + rows_total, n_pix = pix_before + pix_total;
        pointers[num_points] = n_rows;
        if (n_rows <= out_capacity) view_ptr[n_rows] = n_pix;
        counters[0] = n_rows;
        counters[3] = view_rank[V];
        counters[4] = n_pix;
        counters[5] = (i64)mask_cursor[0];           // verdict entries the call needs
        counters[6] = (i64)mask_cursor[1];           // chunks of the pixel pass
        if (n_rows > out_capacity) atomicOr((unsigned long long *)&counters[2], 1ull);
        if (n_pix > pix_capacity) atomicOr((unsigned long long *)&counters[2], 4ull);
    }
    if (q >= num_points) return;
    pointers[q] = k;
    if (rows == 0) return;
    const float px3 = src.xyz[3 * i], py3 = src.xyz[3 * i + 1], pz3 = src.xyz[3 * i + 2];
    for (int wd = 0; wd < words; ++wd) {
        unsigned m = masks[(size_t)wd * num_points + q];
        while (m) {
            const int v = (wd << 5) + __ffs(m) - 1;
            m &= m - 1;
            const unsigned c = cnt[(size_t)v * n + i];
            {
                // for the pixel pass: first pixel of the pair minus the pixels in front of its first entry
                NePair pr = pairs[(size_t)v * n + i];
                const i64 fp = p - (pr.x == NE_NO_MASKS ? 0 : (i64)bits_before[pr.x >> 3]);
                pr.z = (unsigned)fp;
                pr.w = (pr.w & 0xffff0000u) | ((unsigned)(fp >> 32) & 0xffffu);
                pairs[(size_t)v * n + i] = pr;
            }
            if (k < out_capacity) {
                double x, y, fx, fy; float d;
                src.get(i, v, P, x, y, d, fx, fy);
                float f[8];
                write_features(f, A, i, px3, py3, pz3, src.views[v].img_xyz, d, y);
                for (int j = 0; j < A.F; ++j)
                    features[(size_t)k * A.F + j] = __fdiv_rn(repeat_add(f[j], c), (float)c);
                images[k] = view_rank[v];
                view_ptr[k] = p;
            }
            p += c;
            ++k;
        }
    }
}

}  // namespace mvf

using namespace mvf;

// =========================================================================== //
// C ABI                                                                        //
// =========================================================================== //

static int check_params(const mvf_vis_params *P, const char *who)
{
    MVF_REQUIRE(P != nullptr, "%s: null params", who);
    MVF_REQUIRE(P->W >= 1 && P->H >= 1 && P->W <= 65535 && P->H <= 65535,
                "%s: image size %dx%d out of range", who, P->W, P->H);
    MVF_REQUIRE(P->crop_top >= 0 && P->crop_bottom >= 0 && P->crop_top + P->crop_bottom < P->H,
                "%s: crops %d+%d do not fit height %d", who, P->crop_top, P->crop_bottom, P->H);
    MVF_REQUIRE(P->camera >= MVF_CAM_KITTI360_PERSPECTIVE && P->camera <= MVF_CAM_S3DIS_EQUIRECTANGULAR,
                "%s: unknown camera %d", who, P->camera);
    return MVF_OK;
}

extern "C" size_t mvf_camera_projection_workspace_bytes(int64_t n)
{
    return align_up(compact_workspace_elems(n) * sizeof(unsigned));
}

extern "C" int mvf_camera_projection(
    const float *xyz, int64_t n, const mvf_view *h_view, const mvf_vis_params *h_params,
    const uint8_t *mask, int64_t *indices, float *dist, double *x_proj, double *y_proj,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_params(h_params, "camera_projection")) return rc;
    MVF_REQUIRE(n >= 0 && h_view && counters, "camera_projection: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { MVF_HIP(hipMemsetAsync(counters, 0, sizeof(int64_t), st)); return MVF_OK; }
    MVF_REQUIRE(xyz && indices && dist && x_proj && y_proj, "camera_projection: null buffer");
    Arena ar(workspace, workspace_bytes);
    unsigned *blocks = ar.take<unsigned>(compact_workspace_elems(n));
    if (!ar.ok()) { set_error("camera_projection: workspace too small"); return MVF_E_WORKSPACE; }
    ProjPred pred{xyz, *h_view, *h_params, mask};
    ProjEmit emit{xyz, *h_view, *h_params, mask, (i64 *)indices, dist, x_proj, y_proj};
    compact(pred, emit, (i64)n, blocks, (i64 *)counters, (i64)n, (i64 *)nullptr, st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// ------------------------------------------------------------------------- //
// camera_projection without a workspace argument and without any allocation   //
// ------------------------------------------------------------------------- //
// The per-block counts of the compaction live in a device array that is part
// of the library's code object.  Every GPU holds its own copy at its own
// address: the address is asked for once per device (with that device
// current) and kept; host code below only ever uses that pointer, never the
// array's name.  Calls on one GPU share the array (it is rewritten from
// scratch by every call, nothing has to start at zero), so calls that overlap
// on one GPU race.
constexpr i64 NOALLOC_MAX_POINTS = 1ll << 24;                             // 16 Mi points
constexpr size_t NOALLOC_SCRATCH_ELEMS = (size_t)(NOALLOC_MAX_POINTS / 256) + 1;   // 256 KB + 4 B
static __device__ __align__(256) unsigned scratch[NOALLOC_SCRATCH_ELEMS];

constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static unsigned *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Device address of `scratch` on the current device (null + error set on failure).
static unsigned *scratch_on_current_device()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        set_error("camera_projection_noalloc: no current HIP device (or index >= %d)", NOALLOC_MAX_DEVICES);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("camera_projection_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = (unsigned *)p;
    }
    return g_scratch_ptr[dev];
}

extern "C" int64_t mvf_camera_projection_noalloc_max_points(void)
{
    // also the set-up call: resolves the scratch array's address on the current device (when
    // there is one), so that no later call has to do it under a graph capture
    int count = 0;
    if (hipGetDeviceCount(&count) == hipSuccess && count > 0) (void)scratch_on_current_device();
    return NOALLOC_MAX_POINTS;
}

extern "C" int mvf_camera_projection_noalloc(
    const float *xyz, int64_t n, const mvf_view *h_view, const mvf_vis_params *h_params,
    const uint8_t *mask, int64_t *indices, float *dist, double *x_proj, double *y_proj,
    int64_t *counters, void *stream)
{
    if (int rc = check_params(h_params, "camera_projection_noalloc")) return rc;
    MVF_REQUIRE(n >= 0 && h_view && counters, "camera_projection_noalloc: bad arguments");
    MVF_REQUIRE(n <= NOALLOC_MAX_POINTS,
                "camera_projection_noalloc: n=%lld exceeds the built-in scratch (at most %lld points); "
                "use mvf_camera_projection with a workspace", (long long)n, (long long)NOALLOC_MAX_POINTS);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { MVF_HIP(hipMemsetAsync(counters, 0, sizeof(int64_t), st)); return MVF_OK; }
    MVF_REQUIRE(xyz && indices && dist && x_proj && y_proj, "camera_projection_noalloc: null buffer");
    unsigned *blocks = scratch_on_current_device();
    if (!blocks) return MVF_E_LAUNCH;
    ProjPred pred{xyz, *h_view, *h_params, mask};
    ProjEmit emit{xyz, *h_view, *h_params, mask, (i64 *)indices, dist, x_proj, y_proj};
    compact(pred, emit, (i64)n, blocks, (i64 *)counters, (i64)n, (i64 *)nullptr, st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// Persistent z-buffer grid: 3 workgroups of 512 threads (50 KB LDS each) per CU.
static unsigned zb_grid(i64 tiles)
{
    // compute units of each device, asked for once per device
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        int c = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
        cus[dev] = c > 0 ? c : 256;
    }
    const i64 g = (i64)cus[dev] * ZB_WG_PER_CU;
    return (unsigned)(tiles < g ? tiles : g);
}

// Boxes staged by the count pass for the fill pass: 256 slots and one count per
// block of 256 points and view.
static inline size_t staged_bytes(i64 n, int V)
{
    const size_t blocks = (size_t)V * cdiv(n > 0 ? n : 1, 256);
    return align_up(blocks * 256 * sizeof(TileEntry)) + align_up(blocks * sizeof(unsigned));
}

// Bytes of one point's row of per-view flags.
static inline i64 seen_row_bytes(int V) { return ((i64)V + 15) / 16 * 16; }

static inline void tile_grid(const mvf_vis_params &P, int &tiles_x, int &tiles_y, int &Hc)
{
    Hc = P.H - P.crop_top - P.crop_bottom;
    tiles_x = (P.W + TILE_X - 1) / TILE_X;
    tiles_y = (Hc + TILE_Y - 1) / TILE_Y;
}

extern "C" size_t mvf_splat_visibility_workspace_bytes(
    int64_t m, int32_t W, int32_t H, int64_t pair_capacity)
{
    const size_t tiles = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    const size_t npix = (size_t)W * H;
    size_t b = 0;
    b += align_up((tiles * NCLS + BIN_TAIL) * sizeof(unsigned)) * 3;         // count, off, cur
    b += align_up((tiles + 1) * sizeof(TileDesc));
    b += align_up((size_t)(pair_capacity > 0 ? pair_capacity : 1) * sizeof(TileEntry));
    b += staged_bytes(m, 1);
    b += align_up((size_t)(m > 0 ? m : 1));                           // seen
    b += align_up(npix * sizeof(unsigned));                           // idmap
    b += align_up(compact_workspace_elems(npix) * sizeof(unsigned));
    return b;
}

static int splat_visibility_impl(
    const double *x_proj, const double *y_proj, const float *dist, const float *xyz, int64_t m,
    const mvf_view &view, const mvf_vis_params *h_params, int64_t pair_capacity,
    int64_t *out_idx, int64_t *out_x, int64_t *out_y, int64_t out_capacity,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_params(h_params, "splat_visibility")) return rc;
    MVF_REQUIRE(m >= 1 && m < (1ll << 32), "splat_visibility: m=%lld out of range", (long long)m);
    MVF_REQUIRE(x_proj && y_proj && dist && out_idx && out_x && out_y && counters,
                "splat_visibility: null buffer");
    MVF_REQUIRE(pair_capacity >= 1 && out_capacity >= 0, "splat_visibility: bad capacity");
    MVF_REQUIRE(xyz || h_params->camera != MVF_CAM_KITTI360_FISHEYE,
                "splat_visibility: the fisheye splat needs xyz");
    const mvf_vis_params P = *h_params;
    hipStream_t st = (hipStream_t)stream;
    int tiles_x, tiles_y, Hc;
    tile_grid(P, tiles_x, tiles_y, Hc);
    const i64 tiles = (i64)tiles_x * tiles_y;
    const i64 npix = (i64)P.W * Hc;

    Arena ar(workspace, workspace_bytes);
    // counters are [tile][class]; two extra words behind tile_count: number of
    // non-empty tiles, z-buffer ticket counter
    const i64 ncnt = tiles * NCLS;
    unsigned *tile_count = ar.take<unsigned>(ncnt + BIN_TAIL);
    unsigned *tile_cur = ar.take<unsigned>(ncnt + BIN_TAIL);
    unsigned *tile_off = ar.take<unsigned>(ncnt + BIN_TAIL);
    TileDesc *descs = ar.take<TileDesc>(tiles + 1);
    TileEntry *entries = ar.take<TileEntry>(pair_capacity);
    TileEntry *staged = ar.take<TileEntry>((size_t)cdiv(m, 256) * 256);
    unsigned *staged_count = ar.take<unsigned>(cdiv(m, 256));
    uint8_t *seen = ar.take<uint8_t>(m);
    unsigned *idmap = ar.take<unsigned>(npix);
    unsigned *cblocks = ar.take<unsigned>(compact_workspace_elems(npix));
    if (!ar.ok()) { set_error("splat_visibility: workspace too small (%zu < %zu)", workspace_bytes, ar.off); return MVF_E_WORKSPACE; }

    // tile_count and tile_cur are adjacent: one memset; counters too
    MVF_HIP(hipMemsetAsync(tile_count, 0, (size_t)(ncnt + BIN_TAIL) * sizeof(unsigned), st));
    MVF_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int64_t), st));
    if (P.exact) MVF_HIP(hipMemsetAsync(seen, 0, (size_t)m, st));
    MVF_HIP(hipMemsetAsync(idmap, 0, (size_t)npix * sizeof(unsigned), st));

    ArraySrc src{x_proj, y_proj, dist, xyz, view, (i64)m};
    dim3 grid_m(cdiv(m, 256), 1), block(256);
    hipLaunchKernelGGL((bin_count_kernel<ArraySrc, false, BIN_LDS_WIDE>), grid_m, block, 0, st, src, P, tiles_y,
                       (int)tiles, tile_count, (uint8_t *)nullptr, (i64)0, (i64 *)nullptr, (int *)nullptr, 0,
                       staged, staged_count);
    // tile_count[tiles] doubles as the number of non-empty tiles
    hipLaunchKernelGGL(tile_scan_kernel, dim3(cdiv(tiles, 1024)), dim3(1024), 0, st, tile_count, tiles, tile_off, tile_cur,
                       descs, tile_count + ncnt, (i64 *)counters + 1, tiles_y, (int)tiles);
    hipLaunchKernelGGL((bin_fill_kernel<BIN_LDS_WIDE>), grid_m, block, 0, st, staged, staged_count, tiles_y,
                       (int)tiles, tile_off, tile_cur, entries, (i64)pair_capacity, (i64 *)counters + 2,
                       (const int2 *)nullptr);
    if (P.exact) {
        hipLaunchKernelGGL((tile_zbuffer_kernel<0>), dim3(zb_grid(tiles)), dim3(ZB_THREADS), 0, st, entries,
                           descs, tile_count + ncnt, tile_count + ncnt + ZB_QUEUE_STRIDE, (i64)pair_capacity, tiles_y, (int)tiles, P.W, Hc,
                           seen, (i64)1, (unsigned *)nullptr, (const int *)nullptr);
        hipLaunchKernelGGL(centre_stamp_kernel, grid_m, block, 0, st, x_proj, y_proj, (i64)m, seen,
                           P.crop_top, Hc, idmap);
    } else {
        hipLaunchKernelGGL((tile_zbuffer_kernel<1>), dim3(zb_grid(tiles)), dim3(ZB_THREADS), 0, st, entries,
                           descs, tile_count + ncnt, tile_count + ncnt + ZB_QUEUE_STRIDE, (i64)pair_capacity, tiles_y, (int)tiles, P.W, Hc,
                           (uint8_t *)nullptr, (i64)0, idmap, (const int *)nullptr);
    }
    IdmapPred pred{idmap};
    IdmapEmit emit{idmap, Hc, P.crop_top, (i64 *)out_idx, (i64 *)out_x, (i64 *)out_y};
    compact(pred, emit, npix, cblocks, (i64 *)counters, (i64)out_capacity, (i64 *)counters + 2, st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_splat_visibility(
    const double *x_proj, const double *y_proj, const float *dist, int64_t m,
    float fx, float fy, const mvf_vis_params *h_params, int64_t pair_capacity,
    int64_t *out_idx, int64_t *out_x, int64_t *out_y, int64_t out_capacity,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(h_params && (h_params->camera == MVF_CAM_KITTI360_PERSPECTIVE ||
                             h_params->camera == MVF_CAM_SCANNET),
                "splat_visibility: pinhole cameras only (use mvf_splat_visibility_view)");
    mvf_view view = {};
    view.fx = fx; view.fy = fy;
    return splat_visibility_impl(x_proj, y_proj, dist, nullptr, m, view, h_params, pair_capacity,
                                 out_idx, out_x, out_y, out_capacity, counters, workspace,
                                 workspace_bytes, stream);
}

extern "C" int mvf_splat_visibility_view(
    const double *x_proj, const double *y_proj, const float *dist, const float *xyz, int64_t m,
    const mvf_view *h_view, const mvf_vis_params *h_params, int64_t pair_capacity,
    int64_t *out_idx, int64_t *out_x, int64_t *out_y, int64_t out_capacity,
    int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(h_view != nullptr, "splat_visibility_view: null view");
    return splat_visibility_impl(x_proj, y_proj, dist, xyz, m, *h_view, h_params, pair_capacity,
                                 out_idx, out_x, out_y, out_capacity, counters, workspace,
                                 workspace_bytes, stream);
}

extern "C" size_t mvf_depth_map_visibility_workspace_bytes(int64_t m)
{
    return align_up(compact_workspace_elems(m) * sizeof(unsigned));
}

extern "C" int mvf_depth_map_visibility(
    const double *x_proj, const double *y_proj, const float *dist, int64_t m,
    const float *depth_map, int32_t W, int32_t H, float depth_threshold,
    int64_t *out_idx, double *out_x, double *out_y, int64_t *counters,
    void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(m >= 1 && W >= 1 && H >= 1, "depth_map_visibility: bad shape");
    MVF_REQUIRE(x_proj && y_proj && dist && depth_map && out_idx && out_x && out_y && counters,
                "depth_map_visibility: null buffer");
    Arena ar(workspace, workspace_bytes);
    unsigned *blocks = ar.take<unsigned>(compact_workspace_elems(m));
    if (!ar.ok()) { set_error("depth_map_visibility: workspace too small"); return MVF_E_WORKSPACE; }
    DepthPred pred{x_proj, y_proj, dist, depth_map, H, depth_threshold};
    DepthEmit emit{x_proj, y_proj, (i64 *)out_idx, out_x, out_y};
    compact(pred, emit, (i64)m, blocks, (i64 *)counters, (i64)m, (i64 *)nullptr, (hipStream_t)stream);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

static FeatArgs feat_args(const float *lin, const float *pla, const float *sca, const float *nrm,
                          int H, double r_max, double r_min)
{
    FeatArgs A;
    A.lin = lin; A.pla = pla; A.sca = sca; A.nrm = nrm;
    A.F = 2 + (lin != nullptr) + (pla != nullptr) + (sca != nullptr) + (nrm != nullptr);
    A.H = H; A.lo = (float)r_min; A.hi = (float)(r_max + 1e-4);
    return A;
}

extern "C" int mvf_mapping_features(
    const float *xyz, const int64_t *idx, const double *y_proj, const float *dist, int64_t k,
    const float *h_img_xyz, const float *linearity, const float *planarity,
    const float *scattering, const float *normals, int32_t H, double r_max, double r_min,
    float *out, void *stream)
{
    MVF_REQUIRE(k >= 0 && H >= 1 && h_img_xyz, "mapping_features: bad arguments");
    if (k == 0) return MVF_OK;
    MVF_REQUIRE(xyz && idx && y_proj && dist && out, "mapping_features: null buffer");
    Float3 c{{h_img_xyz[0], h_img_xyz[1], h_img_xyz[2]}};
    hipLaunchKernelGGL(features_kernel, dim3(cdiv(k, 256)), dim3(256), 0, (hipStream_t)stream, xyz,
                       (const i64 *)idx, y_proj, dist, (i64)k, c,
                       feat_args(linearity, planarity, scattering, normals, H, r_max, r_min), out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// Number of slots of the centre-pixel table for a requested size: a power of
// two, at least 1024.
static inline i64 centre_table_slots(i64 requested);

// Entries of the verdict buffer of the plain-splat pass: the caller's number (0 = half
// an entry, that is four bits, per pixel of the views: C2 needs 0.44), a multiple of 64
// below 2^32.
static inline i64 ne_mask_capacity(i64 requested, i64 map_pixels)
{
    i64 c = requested > 0 ? requested : map_pixels / 2;
    c = (c + 63) & ~(i64)63;
    if (c < 64) c = 64;
    if (c > 0xfffffff0ll) c = 0xfffffff0ll;
    return c;
}

// The buffers mvf_map_images_exact cuts out of the caller's workspace.
struct MapWorkspace {
    i64 ncnt, seen_stride;
    unsigned *tile_count, *tile_cur, *tile_off, *staged_count, *masks, *block_sums;
    TileDesc *descs;
    TileEntry *entries, *staged;
    uint8_t *seen;
    CentreTable table;
    size_t clear_bytes, used;
    // plain splat visibility only (map_pixels > 0): id map [V][W][Hc], pixel counts of the
    // (point, view) pairs [V][n], pixels per block, and the verdict bits (one byte per entry)
    // with their running counts per 8 entries, the pair records [V][n] and the chunk list
    // (see ne_box_kernel)
    unsigned *idmap, *cnt, *bits_blocks, *bits_before;
    i64 *block_pix;
    uint8_t *verdicts;
    uint4 *ne_pairs, *ne_chunks;
    i64 chunk_capacity;
    i64 mask_capacity;
    // Vw: views a point can meet (V, or the largest sample of a batched call)
    bool carve(void *workspace, size_t workspace_bytes, i64 n, int V, i64 tiles, i64 num_points,
               i64 pair_capacity, i64 centre_slots, i64 map_pixels = 0, int Vw = 0, i64 mask_entries = 0)
    {
        if (Vw <= 0) Vw = V;
        const i64 slots = centre_table_slots(centre_slots);
        int slots_log = 0;
        while ((1ll << slots_log) < slots) ++slots_log;
        const int words = (Vw + 31) / 32;
        Arena ar(workspace, workspace_bytes);
        // counters are [tile][class]; behind them: number of non-empty tiles, z-buffer tickets
        ncnt = tiles * NCLS;
        tile_count = ar.take<unsigned>(ncnt + BIN_TAIL);
        // the centre table sits right behind the counters: one memset clears both
        table = CentreTable{ar.take<u64>((size_t)slots), (unsigned)(slots - 1), 64 - slots_log};
        clear_bytes = ar.off;
        tile_cur = ar.take<unsigned>(ncnt + BIN_TAIL);
        tile_off = ar.take<unsigned>(ncnt + BIN_TAIL);
        descs = ar.take<TileDesc>(tiles + 1);
        entries = ar.take<TileEntry>(pair_capacity);
        staged = ar.take<TileEntry>((size_t)Vw * cdiv(n, 256) * 256);
        staged_count = ar.take<unsigned>((size_t)Vw * cdiv(n, 256));
        seen_stride = seen_row_bytes(Vw);
        seen = ar.take<uint8_t>((size_t)seen_stride * n);
        masks = ar.take<unsigned>((size_t)words * num_points);
        block_sums = ar.take<unsigned>(cdiv(num_points, 256));
        idmap = cnt = bits_blocks = bits_before = nullptr;
        block_pix = nullptr;
        verdicts = nullptr;
        ne_pairs = ne_chunks = nullptr;
        mask_capacity = chunk_capacity = 0;
        if (map_pixels > 0) {
            idmap = ar.take<unsigned>((size_t)map_pixels);
            cnt = ar.take<unsigned>((size_t)n * V);
            block_pix = ar.take<i64>(cdiv(num_points, 256));
            mask_capacity = ne_mask_capacity(mask_entries, map_pixels);
            verdicts = ar.take<uint8_t>((size_t)mask_capacity + 64);
            bits_before = ar.take<unsigned>((size_t)mask_capacity / 8 + 16);
            bits_blocks = ar.take<unsigned>((size_t)cdiv(mask_capacity / 8 + 1, 2048) + 1);
            ne_pairs = ar.take<uint4>((size_t)n * V);
            // a seen pair has at least one chunk and at least one (splat, tile) entry; then one
            // chunk per NE_CHUNK words
            chunk_capacity = pair_capacity + mask_capacity / (8 * NE_CHUNK) + 1;
            ne_chunks = ar.take<uint4>((size_t)chunk_capacity);
        }
        used = ar.off;
        return ar.ok();
    }
};

static inline i64 centre_table_slots(i64 requested)
{
    i64 t = 1024;
    while (t < requested && t < (1ll << 31)) t <<= 1;
    return t;
}

extern "C" size_t mvf_map_images_workspace_bytes(
    int64_t n, int32_t V, int32_t W, int32_t H, int64_t num_points, int64_t pair_capacity,
    int64_t centre_slots)
{
    const i64 tiles = (i64)V * ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    MapWorkspace ws;
    ws.carve(nullptr, 0, n > 0 ? n : 1, V, tiles, num_points > 0 ? num_points : 1,
             pair_capacity > 0 ? pair_capacity : 1, centre_slots);
    return ws.used;
}

// LDS-atomic work of the z-buffer pass of the mvf_map_images_exact call that
// last used `workspace` (same sizes): d_out[0..2] += wave-instructions, lane
// atomics, entries (see zbuffer_count_kernel); d_out[3] += in-view (point, view)
// pairs (the records the count pass staged).
extern "C" int mvf_profile_zbuffer_ops(
    void *workspace, size_t workspace_bytes, int64_t n, int32_t V, const mvf_vis_params *h_params,
    int64_t num_points, int64_t pair_capacity, int64_t centre_slots, uint64_t *d_out, void *stream)
{
    if (int rc = check_params(h_params, "profile_zbuffer_ops")) return rc;
    MVF_REQUIRE(workspace && d_out && n >= 1 && V >= 1, "profile_zbuffer_ops: bad arguments");
    int tiles_x, tiles_y, Hc;
    tile_grid(*h_params, tiles_x, tiles_y, Hc);
    const i64 tiles = (i64)tiles_x * tiles_y * V;
    MapWorkspace ws;
    if (!ws.carve(workspace, workspace_bytes, n, V, tiles, num_points, pair_capacity, centre_slots)) {
        set_error("profile_zbuffer_ops: workspace too small");
        return MVF_E_WORKSPACE;
    }
    hipLaunchKernelGGL(zbuffer_count_kernel, dim3(zb_grid(tiles)), dim3(ZB_THREADS), 0, (hipStream_t)stream,
                       ws.entries, ws.descs, ws.tile_count + ws.ncnt, (i64)pair_capacity, (u64 *)d_out,
                       (unsigned)tiles);
    const i64 staged_blocks = (i64)V * cdiv(n, 256);
    hipLaunchKernelGGL(staged_total_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, ws.staged_count,
                       staged_blocks, (u64 *)d_out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// The fused exact pass.  window / view_first / Vw describe a batch of samples
// (null, null, V: one sample whose points meet all V views).
template <class Src>
static int map_images_exact_impl(
    const float *xyz, const int64_t *inv_ids, int64_t n,
    int64_t num_points, const mvf_view *views, int32_t V, const mvf_vis_params *h_params,
    int32_t cylinder, const uint8_t *mask, const float *linearity, const float *planarity,
    const float *scattering, const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t centre_slots, int64_t pair_capacity,
    int64_t *pointers, int64_t *images, int16_t *pixels, float *features, int64_t out_capacity,
    int32_t *view_seen, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream,
    const int2 *window, const int *view_first, int32_t Vw)
{
    if (int rc = check_params(h_params, "map_images_exact")) return rc;
    MVF_REQUIRE(Vw >= 1 && Vw <= V, "map_images_exact: bad views per point %d (V = %d)", Vw, V);
    MVF_REQUIRE(n >= 1 && n < (1ll << 32) - 1 && num_points >= 1 && V >= 1 && V <= MI_MAX_VIEWS,
                "map_images_exact: bad sizes n=%lld V=%d (at most %d views per call)", (long long)n, V, MI_MAX_VIEWS);
    MVF_REQUIRE(xyz && inv_ids && views && pointers && images && pixels && features &&
                view_seen && counters, "map_images_exact: null buffer");
    MVF_REQUIRE(proj_upscale >= 1 && downscale >= 1 && crop_w >= 1 && crop_h >= 1 &&
                pair_capacity >= 1 && out_capacity >= 0 && centre_slots >= 1,
                "map_images_exact: bad scalar argument");
    MVF_REQUIRE(h_params->exact, "map_images_exact: params.exact must be set");
    MVF_REQUIRE((double)V * h_params->W * h_params->H < 4294967295.0,
                "map_images_exact: V * W * H = %d * %d * %d does not fit 32 bits", V, h_params->W, h_params->H);
    const mvf_vis_params P = *h_params;
    hipStream_t st = (hipStream_t)stream;
    int tiles_x, tiles_y, Hc;
    tile_grid(P, tiles_x, tiles_y, Hc);
    const int tiles_per_view = tiles_x * tiles_y;
    const i64 tiles = (i64)tiles_per_view * V;
    const i64 nb = cdiv(num_points, 256);
    MapWorkspace ws;
    if (!ws.carve(workspace, workspace_bytes, n, V, tiles, num_points, pair_capacity, centre_slots, 0, Vw)) {
        set_error("map_images_exact: workspace too small (%zu < %zu)", workspace_bytes, ws.used);
        return MVF_E_WORKSPACE;
    }
    const i64 ncnt = ws.ncnt, seen_stride = ws.seen_stride;
    unsigned *tile_count = ws.tile_count, *tile_cur = ws.tile_cur, *tile_off = ws.tile_off;
    TileDesc *descs = ws.descs;
    TileEntry *entries = ws.entries, *staged = ws.staged;
    unsigned *staged_count = ws.staged_count, *masks = ws.masks, *block_sums = ws.block_sums;
    uint8_t *seen = ws.seen;
    const CentreTable table = ws.table;
    const size_t clear_bytes = ws.clear_bytes;

    prof_begin(st);
    MVF_HIP(hipMemsetAsync(tile_count, 0, clear_bytes, st));
    prof_mark("memset", st);

    Src src{xyz, views, mask, (i64)n, cylinder, window};
    dim3 grid_nv(cdiv(n, 256), Vw), block(256);
    if (window)
        hipLaunchKernelGGL((bin_count_kernel<Src, true, BIN_LDS, true>), grid_nv, block, 0, st, src, P, tiles_y,
                           tiles_per_view, tile_count, seen, seen_stride, (i64 *)counters, view_seen, V, staged,
                           staged_count);
    else
        hipLaunchKernelGGL((bin_count_kernel<Src, true, BIN_LDS>), grid_nv, block, 0, st, src, P, tiles_y,
                           tiles_per_view, tile_count, seen, seen_stride, (i64 *)counters, view_seen, V, staged,
                           staged_count);
    prof_mark("bin_count", st);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(cdiv(tiles, 1024)), dim3(1024), 0, st, tile_count, tiles, tile_off, tile_cur,
                       descs, tile_count + ncnt, (i64 *)counters + 1, tiles_y, tiles_per_view);
    prof_mark("tile_scan", st);
    if (window)
        hipLaunchKernelGGL((bin_fill_kernel<BIN_LDS, true>), grid_nv, block, 0, st, staged, staged_count, tiles_y,
                           tiles_per_view, tile_off, tile_cur, entries, (i64)pair_capacity, (i64 *)counters + 2, window);
    else
        hipLaunchKernelGGL((bin_fill_kernel<BIN_LDS>), grid_nv, block, 0, st, staged, staged_count, tiles_y,
                           tiles_per_view, tile_off, tile_cur, entries, (i64)pair_capacity, (i64 *)counters + 2, window);
    prof_mark("bin_fill", st);
    hipLaunchKernelGGL((tile_zbuffer_kernel<0>), dim3(zb_grid(tiles)), dim3(ZB_THREADS), 0, st, entries,
                       descs, tile_count + ncnt, tile_count + ncnt + ZB_QUEUE_STRIDE, (i64)pair_capacity, tiles_y, tiles_per_view, P.W, Hc,
                       seen, seen_stride, (unsigned *)nullptr, view_first);
    prof_mark("tile_zbuffer", st);
    hipLaunchKernelGGL((mi_centre_kernel<Src>), dim3(cdiv(n, 256)), block, (size_t)V * sizeof(mvf_view), st, src, P, seen, seen_stride, V, Hc,
                       table, (i64 *)counters + 2);
    prof_mark("centre", st);

    PixelXform X{proj_upscale, downscale, crop_w, crop_h};
    dim3 grid_q((unsigned)nb);
    hipLaunchKernelGGL((mi_point_mask_kernel<SplatKeep, Src>), grid_q, block, (size_t)V * sizeof(mvf_view), st, src, P,
                       SplatKeep{seen, seen_stride, table, Hc}, X, V, Vw, (const i64 *)inv_ids, (i64)num_points,
                       masks, block_sums, view_seen);
    prof_mark("point_mask", st);
    hipLaunchKernelGGL((mi_fill_kernel<Src>), grid_q, block, (size_t)V * sizeof(mvf_view) + (size_t)(V + 1) * sizeof(int), st, src, P, X, V, Vw,
                       (const i64 *)inv_ids, (i64)num_points, masks, block_sums, view_seen,
                       feat_args(linearity, planarity, scattering, normals, P.H, P.r_max, P.r_min),
                       (i64)out_capacity, (i64 *)pointers, (i64 *)images, pixels, features,
                       (i64 *)counters);
    prof_mark("csr_fill", st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_map_images_exact(
    const float *xyz, const int64_t *point_ids, const int64_t *inv_ids, int64_t n,
    int64_t num_points, const mvf_view *views, int32_t V, const mvf_vis_params *h_params,
    int32_t cylinder, const uint8_t *mask, const float *linearity, const float *planarity,
    const float *scattering, const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t centre_slots, int64_t pair_capacity,
    int64_t *pointers, int64_t *images, int16_t *pixels, float *features, int64_t out_capacity,
    int32_t *view_seen, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    (void)point_ids;
    if (h_params && h_params->camera >= MVF_CAM_KITTI360_FISHEYE)
        return map_images_exact_impl<CloudSrcAny>(
            xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering,
            normals, proj_upscale, downscale, crop_w, crop_h, centre_slots, pair_capacity, pointers, images,
            pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream,
            (const int2 *)nullptr, (const int *)nullptr, V);
    return map_images_exact_impl<CloudSrc>(
        xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering,
        normals, proj_upscale, downscale, crop_w, crop_h, centre_slots, pair_capacity, pointers, images,
        pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream,
        (const int2 *)nullptr, (const int *)nullptr, V);
}

// A batch of samples in one launch set: the clouds concatenated, the views
// concatenated, every point facing the views of its own sample only.
extern "C" size_t mvf_map_images_batch_workspace_bytes(
    int64_t n, int32_t V, int32_t views_per_point, int32_t W, int32_t H, int64_t num_points,
    int64_t pair_capacity, int64_t centre_slots)
{
    const i64 tiles = (i64)V * ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    MapWorkspace ws;
    ws.carve(nullptr, 0, n > 0 ? n : 1, V, tiles, num_points > 0 ? num_points : 1,
             pair_capacity > 0 ? pair_capacity : 1, centre_slots, 0, views_per_point);
    return ws.used;
}

extern "C" int mvf_map_images_exact_batch(
    const float *xyz, const int64_t *inv_ids, const int32_t *view_window, int64_t n,
    int64_t num_points, const mvf_view *views, const int32_t *view_first, int32_t V,
    int32_t views_per_point, const mvf_vis_params *h_params,
    int32_t cylinder, const uint8_t *mask, const float *linearity, const float *planarity,
    const float *scattering, const float *normals, int32_t proj_upscale, int32_t downscale,
    int32_t crop_w, int32_t crop_h, int64_t centre_slots, int64_t pair_capacity,
    int64_t *pointers, int64_t *images, int16_t *pixels, float *features, int64_t out_capacity,
    int32_t *view_seen, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    MVF_REQUIRE(view_window && view_first, "map_images_exact_batch: null view window");
    if (h_params && h_params->camera >= MVF_CAM_KITTI360_FISHEYE)
        return map_images_exact_impl<CloudSrcAny>(
            xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering,
            normals, proj_upscale, downscale, crop_w, crop_h, centre_slots, pair_capacity, pointers, images,
            pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream,
            (const int2 *)view_window, (const int *)view_first, views_per_point);
    return map_images_exact_impl<CloudSrc>(
        xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering,
        normals, proj_upscale, downscale, crop_w, crop_h, centre_slots, pair_capacity, pointers, images,
        pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream,
        (const int2 *)view_window, (const int *)view_first, views_per_point);
}

// MapImages with DepthBasedVisibility, all views in one launch set: no z-buffer
// and no shared-pixel rule, so the pass is the CSR assembly alone with the
// depth-map test as the visibility model.
extern "C" size_t mvf_map_images_depth_workspace_bytes(int64_t num_points, int32_t V)
{
    const size_t nb = cdiv(num_points > 0 ? num_points : 1, 256);
    const size_t words = (size_t)(V + 31) / 32;
    return align_up(words * (size_t)(num_points > 0 ? num_points : 1) * sizeof(unsigned)) +
           align_up(nb * sizeof(unsigned));
}

template <class Src>
static int map_images_depth_impl(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points, const mvf_view *views,
    int32_t V, const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *depth_maps, const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t proj_upscale, int32_t downscale, int32_t crop_w, int32_t crop_h,
    int64_t *pointers, int64_t *images, int16_t *pixels, float *features, int64_t out_capacity,
    int32_t *view_seen, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_params(h_params, "map_images_depth")) return rc;
    MVF_REQUIRE(n >= 1 && n < (1ll << 32) - 1 && num_points >= 1 && V >= 1 && V <= MI_MAX_VIEWS,
                "map_images_depth: bad sizes n=%lld V=%d (at most %d views per call)", (long long)n, V, MI_MAX_VIEWS);
    MVF_REQUIRE(xyz && inv_ids && views && depth_maps && pointers && images && pixels && features &&
                view_seen && counters, "map_images_depth: null buffer");
    MVF_REQUIRE(proj_upscale >= 1 && downscale >= 1 && crop_w >= 1 && crop_h >= 1 && out_capacity >= 0,
                "map_images_depth: bad scalar argument");
    const mvf_vis_params P = *h_params;
    hipStream_t st = (hipStream_t)stream;
    const i64 nb = cdiv(num_points, 256);
    const int words = (V + 31) / 32;
    Arena ar(workspace, workspace_bytes);
    unsigned *masks = ar.take<unsigned>((size_t)words * num_points);
    unsigned *block_sums = ar.take<unsigned>(nb);
    if (!ar.ok()) { set_error("map_images_depth: workspace too small (%zu < %zu)", workspace_bytes, ar.off); return MVF_E_WORKSPACE; }

    prof_begin(st);
    MVF_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int64_t), st));
    MVF_HIP(hipMemsetAsync(view_seen, 0, (size_t)V * sizeof(int32_t), st));
    prof_mark("memset", st);
    Src src{xyz, views, mask, (i64)n, cylinder};
    PixelXform X{proj_upscale, downscale, crop_w, crop_h};
    dim3 grid_q((unsigned)nb), block(256);
    hipLaunchKernelGGL((mi_point_mask_kernel<DepthKeep, Src>), grid_q, block, (size_t)V * sizeof(mvf_view), st, src, P,
                       DepthKeep{depth_maps}, X, V, V, (const i64 *)inv_ids, (i64)num_points, masks, block_sums,
                       view_seen);
    prof_mark("point_mask", st);
    hipLaunchKernelGGL((mi_fill_kernel<Src>), grid_q, block, (size_t)V * sizeof(mvf_view) + (size_t)(V + 1) * sizeof(int), st, src, P, X, V, V,
                       (const i64 *)inv_ids, (i64)num_points, masks, block_sums, view_seen,
                       feat_args(linearity, planarity, scattering, normals, P.H, P.r_max, P.r_min),
                       (i64)out_capacity, (i64 *)pointers, (i64 *)images, pixels, features,
                       (i64 *)counters);
    prof_mark("csr_fill", st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_map_images_depth(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points, const mvf_view *views,
    int32_t V, const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *depth_maps, const float *linearity, const float *planarity, const float *scattering,
    const float *normals, int32_t proj_upscale, int32_t downscale, int32_t crop_w, int32_t crop_h,
    int64_t *pointers, int64_t *images, int16_t *pixels, float *features, int64_t out_capacity,
    int32_t *view_seen, int64_t *counters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (h_params && h_params->camera >= MVF_CAM_KITTI360_FISHEYE)
        return map_images_depth_impl<CloudSrcAny>(
            xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, depth_maps, linearity, planarity, scattering, normals, proj_upscale, downscale, crop_w, crop_h, pointers, images, pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream);
    return map_images_depth_impl<CloudSrc>(
        xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, depth_maps, linearity, planarity, scattering, normals, proj_upscale, downscale, crop_w, crop_h, pointers, images, pixels, features, out_capacity, view_seen, counters, workspace, workspace_bytes, stream);
}

// MapImages with SplattingVisibility(exact=False), all views in one launch set:
// binning and tile z-buffer as in the exact pass, then every pixel a point wins
// becomes a mapping (no shared-pixel rule, no centre table).
extern "C" size_t mvf_map_images_splat_workspace_bytes(
    int64_t n, int32_t V, int32_t W, int32_t H, int64_t num_points, int64_t pair_capacity,
    int64_t mask_capacity)
{
    const i64 tiles = (i64)V * ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    MapWorkspace ws;
    ws.carve(nullptr, 0, n > 0 ? n : 1, V, tiles, num_points > 0 ? num_points : 1,
             pair_capacity > 0 ? pair_capacity : 1, 1, (i64)V * W * H, 0, mask_capacity);
    return ws.used;
}

template <class Src>
static int map_images_splat_impl(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points, const mvf_view *views,
    int32_t V, const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *linearity, const float *planarity, const float *scattering, const float *normals,
    int32_t crop_w, int32_t crop_h, int64_t pair_capacity, int64_t mask_capacity, int64_t *pointers,
    int64_t *images, int64_t *view_pointers, int16_t *pixels, float *features, int64_t out_capacity,
    int64_t pix_capacity, int32_t *view_seen, int64_t *counters, void *workspace,
    size_t workspace_bytes, void *stream)
{
    if (int rc = check_params(h_params, "map_images_splat")) return rc;
    MVF_REQUIRE(n >= 1 && n < (1ll << 32) - 1 && num_points >= 1 && V >= 1 && V <= MI_MAX_VIEWS,
                "map_images_splat: bad sizes n=%lld V=%d (at most %d views per call)", (long long)n, V, MI_MAX_VIEWS);
    MVF_REQUIRE(xyz && inv_ids && views && pointers && images && view_pointers && pixels && features &&
                view_seen && counters, "map_images_splat: null buffer");
    MVF_REQUIRE(crop_w >= 1 && crop_h >= 1 && pair_capacity >= 1 && out_capacity >= 0 && pix_capacity >= 0 &&
                mask_capacity >= 0, "map_images_splat: bad scalar argument");
    MVF_REQUIRE((i64)n * V < 0xffffffffll, "map_images_splat: %lld points x %d views are too many for one call", (long long)n, V);
    MVF_REQUIRE((i64)V * h_params->W * h_params->H < 0xffffffffll, "map_images_splat: %d views of %dx%d pixels are too many for one call",
                V, h_params->W, h_params->H);
    const mvf_vis_params P = *h_params;
    hipStream_t st = (hipStream_t)stream;
    int tiles_x, tiles_y, Hc;
    tile_grid(P, tiles_x, tiles_y, Hc);
    const int tiles_per_view = tiles_x * tiles_y;
    const i64 tiles = (i64)tiles_per_view * V;
    const i64 nb = cdiv(num_points, 256);
    MapWorkspace ws;
    if (!ws.carve(workspace, workspace_bytes, n, V, tiles, num_points, pair_capacity, 1, (i64)V * P.W * Hc, 0,
                  mask_capacity > 0 ? mask_capacity : ne_mask_capacity(0, (i64)V * P.W * P.H))) {
        set_error("map_images_splat: workspace too small (%zu < %zu)", workspace_bytes, ws.used);
        return MVF_E_WORKSPACE;
    }
    const i64 ncnt = ws.ncnt;

    prof_begin(st);
    MVF_HIP(hipMemsetAsync(ws.tile_count, 0, (size_t)(ncnt + BIN_TAIL) * sizeof(unsigned), st));
    MVF_HIP(hipMemsetAsync(ws.verdicts, 0, (size_t)ws.mask_capacity + 64, st));
    prof_mark("memset", st);
    Src src{xyz, views, mask, (i64)n, cylinder};
    dim3 grid_nv(cdiv(n, 256), V), block(256);
    hipLaunchKernelGGL((bin_count_kernel<Src, true, BIN_LDS>), grid_nv, block, 0, st, src, P, tiles_y,
                       tiles_per_view, ws.tile_count, ws.seen, ws.seen_stride, (i64 *)counters, view_seen, V,
                       ws.staged, ws.staged_count);
    prof_mark("bin_count", st);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(cdiv(tiles, 1024)), dim3(1024), 0, st, ws.tile_count, tiles,
                       ws.tile_off, ws.tile_cur, ws.descs, ws.tile_count + ncnt, (i64 *)counters + 1, tiles_y,
                       tiles_per_view);
    prof_mark("tile_scan", st);
    hipLaunchKernelGGL((bin_fill_kernel<BIN_LDS>), grid_nv, block, 0, st, ws.staged, ws.staged_count, tiles_y,
                       tiles_per_view, ws.tile_off, ws.tile_cur, ws.entries, (i64)pair_capacity,
                       (i64 *)counters + 2, (const int2 *)nullptr);
    prof_mark("bin_fill", st);
    hipLaunchKernelGGL((tile_zbuffer_kernel<2>), dim3(zb_grid(tiles)), dim3(ZB_THREADS), 0, st, ws.entries,
                       ws.descs, ws.tile_count + ncnt, ws.tile_count + ncnt + ZB_QUEUE_STRIDE, (i64)pair_capacity,
                       tiles_y, tiles_per_view, P.W, Hc, ws.seen, ws.seen_stride, ws.idmap, (const int *)nullptr);
    prof_mark("tile_zbuffer", st);
    u64 *cursor = (u64 *)(ws.tile_count + ncnt + NE_CURSOR_SLOT);        // [0] entries, [1] chunks
    const uint8_t *cseen = ws.seen;
    const unsigned *n_desc = ws.tile_count + ncnt;
    const u64 *verdicts = (const u64 *)ws.verdicts;
    const NeImage img{ws.idmap, P.W, Hc, P.crop_top, crop_w, crop_h, (i64)n};
    hipLaunchKernelGGL(ne_box_kernel, dim3(cdiv(grid_nv.x, NE_BOX_GROUP), V), block, 0, st, ws.staged, ws.staged_count,
                       grid_nv.x, cseen, ws.seen_stride, (i64)n, ws.ne_pairs, ws.ne_chunks, ws.chunk_capacity,
                       ws.mask_capacity, cursor, (i64 *)counters);
    prof_mark("pixel_boxes", st);
    hipLaunchKernelGGL(ne_mark_kernel, dim3((unsigned)(tiles * NE_MARK_BLOCKS)), block, 0, st, ws.descs, n_desc, img,
                       views, (const NePair *)ws.ne_pairs, ws.verdicts);
    prof_mark("pixel_mark", st);
    const unsigned scan_blocks = (unsigned)cdiv(ws.mask_capacity / 8 + 1, NE_SCAN_WORDS);
    hipLaunchKernelGGL(ne_bits_reduce_kernel, dim3(scan_blocks), block, 0, st, verdicts, cursor, ws.mask_capacity,
                       ws.bits_blocks);
    hipLaunchKernelGGL((scan_spine_kernel<unsigned, i64>), dim3(1), dim3(SCAN_BLOCK), 0, st, ws.bits_blocks,
                       (i64)scan_blocks, (i64 *)nullptr);
    hipLaunchKernelGGL(ne_bits_apply_kernel, dim3(scan_blocks), block, 0, st, verdicts, cursor, ws.mask_capacity,
                       (const unsigned *)ws.bits_blocks, ws.bits_before);
    prof_mark("pixel_scan", st);
    hipLaunchKernelGGL(ne_count_kernel, grid_nv, block, 0, st, ws.staged, ws.staged_count, cseen, ws.seen_stride,
                       (i64)n, (const NePair *)ws.ne_pairs, (const unsigned *)ws.bits_before, ws.cnt);
    prof_mark("pixel_count", st);
    dim3 grid_q((unsigned)nb);
    hipLaunchKernelGGL(ne_point_kernel, grid_q, block, 0, st, ws.seen, ws.seen_stride, ws.cnt, (i64)n, V,
                       (const i64 *)inv_ids, (i64)num_points, ws.masks, ws.block_sums, ws.block_pix, view_seen);
    prof_mark("point_mask", st);
    hipLaunchKernelGGL((ne_fill_kernel<Src>), grid_q, block, (size_t)V * sizeof(mvf_view) + (size_t)(V + 1) * sizeof(int), st,
                       src, P, V, (const i64 *)inv_ids, (i64)num_points, ws.masks, ws.block_sums, ws.block_pix,
                       ws.cnt, view_seen, feat_args(linearity, planarity, scattering, normals, P.H, P.r_max, P.r_min),
                       (i64)out_capacity, (i64)pix_capacity, (i64 *)pointers, (i64 *)images, (i64 *)view_pointers,
                       features, ws.ne_pairs, (const unsigned *)ws.bits_before, cursor, (i64 *)counters);
    prof_mark("csr_fill", st);
    // a fixed grid that fills the device; the waves stride over the chunks
    hipLaunchKernelGGL(ne_emit_kernel, dim3(256 * 8), block, 0, st, (const uint4 *)ws.ne_chunks, cursor,
                       (const NePair *)ws.ne_pairs, verdicts, (const unsigned *)ws.bits_before, views, P.crop_top,
                       (unsigned *)pixels, (i64)pix_capacity, ws.chunk_capacity);
    prof_mark("pixel_fill", st);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

extern "C" int mvf_map_images_splat(
    const float *xyz, const int64_t *inv_ids, int64_t n, int64_t num_points, const mvf_view *views,
    int32_t V, const mvf_vis_params *h_params, int32_t cylinder, const uint8_t *mask,
    const float *linearity, const float *planarity, const float *scattering, const float *normals,
    int32_t crop_w, int32_t crop_h, int64_t pair_capacity, int64_t mask_capacity, int64_t *pointers,
    int64_t *images, int64_t *view_pointers, int16_t *pixels, float *features, int64_t out_capacity,
    int64_t pix_capacity, int32_t *view_seen, int64_t *counters, void *workspace,
    size_t workspace_bytes, void *stream)
{
    if (h_params && h_params->camera >= MVF_CAM_KITTI360_FISHEYE)
        return map_images_splat_impl<CloudSrcAny>(
            xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering, normals, crop_w, crop_h, pair_capacity, mask_capacity, pointers, images, view_pointers, pixels, features, out_capacity, pix_capacity, view_seen, counters, workspace, workspace_bytes, stream);
    return map_images_splat_impl<CloudSrc>(
        xyz, inv_ids, n, num_points, views, V, h_params, cylinder, mask, linearity, planarity, scattering, normals, crop_w, crop_h, pair_capacity, mask_capacity, pointers, images, view_pointers, pixels, features, out_capacity, pix_capacity, view_seen, counters, workspace, workspace_bytes, stream);
}
