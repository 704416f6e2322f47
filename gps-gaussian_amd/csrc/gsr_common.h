// gsr_common.h -- shared device-side types and the workspace layout of the gfx950 rasteriser.
//
// Data layout in HBM (all carved from ONE caller-owned workspace, 256-byte aligned sections):
//   header        GsrHeader (64 B)                       R needed, overflow flag, stats
//   splats[P]     48-byte records {x,y,A,B | C,op,r,g | b,depth,rect_lo,rect_hi}: everything the compositing
//                 kernels gather per instance sits in one record (1-2 cache lines per gather instead of 3 arrays)
//   tile_count[T], tile_offset[T+1], tile_cursor[T]      u32 per 16x16 tile
//   keys[cap]     u64 (depth_bits << 32 | gaussian id), binned per tile, sorted in LDS per tile
//   point_list[cap] u32 sorted gaussian ids (what the compositing kernels walk)
//   final_T[H*W], n_contrib[H*W]                         per-pixel state kept for the backward
//   gacc[P]       48-byte records of backward partial sums {dcolor rgb, dmean2D xy, dconic xx xy yy, dopacity}
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpsgs.h"

#define GSR_TILE 16
#define GSR_TILE_PIX 256

struct __attribute__((aligned(16))) GsrSplat {
    float x, y, A, B;          // pixel-space mean, conic xx, xy
    float C, op, r, g;         // conic yy, opacity, colour
    float b, depth;            // colour, view-space depth
    uint32_t rect_lo, rect_hi; // minx | miny<<16 , maxx | maxy<<16  (tile units; empty rect = culled)
};
static_assert(sizeof(GsrSplat) == 48, "splat record must be 48 bytes");

struct __attribute__((aligned(16))) GsrGradAcc {
    float dr, dg, db, dmx; // dL/dcolor, dL/dmean2D.x (NDC-scaled)
    float dmy, cxx, cxy, cyy; // dL/dmean2D.y, dL/dconic (xy holds HALF the true off-diagonal gradient, like upstream)
    float dop, pad0, pad1, pad2;
};
static_assert(sizeof(GsrGradAcc) == 48, "grad record must be 48 bytes");

struct GsrLayout {
    size_t header, splats, tile_count, tile_offset, tile_cursor, keys, point_list, final_T, n_contrib, gacc, total;
    int gx, gy, T;
};

static inline size_t gsr_align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static inline GsrLayout gsr_layout(int P, int W, int H, int64_t cap) {
    GsrLayout L;
    L.gx = (W + GSR_TILE - 1) / GSR_TILE;
    L.gy = (H + GSR_TILE - 1) / GSR_TILE;
    L.T = L.gx * L.gy;
    size_t o = 0;
    const size_t p = (size_t)(P > 0 ? P : 1), t = (size_t)(L.T > 0 ? L.T : 1), c = (size_t)(cap > 0 ? cap : 1);
    const size_t npix = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1);
    L.header = o;      o = gsr_align_up(o + sizeof(GsrHeader));   // header + tile_count are zeroed by ONE memset
    L.tile_count = o;  o = gsr_align_up(o + t * 4);
    L.tile_offset = o; o = gsr_align_up(o + (t + 1) * 4);
    L.tile_cursor = o; o = gsr_align_up(o + t * 4);
    L.splats = o;      o = gsr_align_up(o + p * sizeof(GsrSplat));
    L.keys = o;        o = gsr_align_up(o + c * 8);
    L.point_list = o;  o = gsr_align_up(o + c * 4);
    L.final_T = o;     o = gsr_align_up(o + npix * 4);
    L.n_contrib = o;   o = gsr_align_up(o + npix * 4);
    L.gacc = o;        o = gsr_align_up(o + p * sizeof(GsrGradAcc));
    L.total = o;
    return L;
}

// Launchers implemented in the .hip files (host side).  All enqueue on `s` and never synchronise.
struct GsrFwdParams {
    int P, W, H, gx, gy;
    const float *means3D, *colors, *opacities, *scales, *rotations;
    float scale_modifier, tanfovx, tanfovy;
    const float *view, *proj, *bg;
    float *out_color;
    int *radii;
    int64_t cap;
};

void gsr_launch_preprocess(const GsrFwdParams &p, GsrSplat *splats, uint32_t *tile_count, GsrHeader *hdr, hipStream_t s);
void gsr_launch_scan(uint32_t *tile_count, uint32_t *tile_offset, uint32_t *tile_cursor, int T, int64_t cap, GsrHeader *hdr, hipStream_t s);
void gsr_launch_scatter(int P, int gx, const GsrSplat *splats, uint32_t *tile_cursor, uint64_t *keys, int64_t cap, const GsrHeader *hdr, hipStream_t s);
void gsr_launch_sort(int T, const uint32_t *tile_offset, uint64_t *keys, uint32_t *point_list, const GsrHeader *hdr, hipStream_t s);
void gsr_launch_composite_fwd(int W, int H, int gx, int gy, const GsrSplat *splats, const uint32_t *tile_offset, const uint32_t *point_list,
                              const float *bg, float *out_color, float *final_T, uint32_t *n_contrib, const GsrHeader *hdr, hipStream_t s);
void gsr_launch_composite_bwd(int W, int H, int gx, int gy, const GsrSplat *splats, const uint32_t *tile_offset, const uint32_t *point_list,
                              const float *bg, const float *dL_dpix, const float *final_T, const uint32_t *n_contrib, GsrGradAcc *gacc,
                              const GsrHeader *hdr, hipStream_t s);
struct GsrBwdParams {
    int P, W, H;
    const float *means3D, *scales, *rotations;
    float scale_modifier, tanfovx, tanfovy;
    const float *view, *proj;
    const int *radii;
    float *dL_dmeans3D, *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dscales, *dL_drotations;
};
void gsr_launch_preprocess_bwd(const GsrBwdParams &p, const GsrGradAcc *gacc, hipStream_t s);
