// gsr_common.h -- shared device-side types and the workspace layout of the gfx950 rasteriser.
//
// Work decomposition (MI355X-first): the image is cut into 8x8-pixel BINS, one wave64 per bin (lane = pixel).  A
// Gaussian is listed in a bin iff the bin lies inside upstream's 16x16-tile rect (the semantic membership rule,
// SURVEY.md section 9.4.1) AND inside the axis-aligned bounding box of its alpha >= 1/255 level set (pairs outside that box
// are guaranteed to fail the alpha test, so dropping them is exact).  Compared with one 256-thread workgroup per
// 16x16 tile this gives ~2.6x shorter per-pixel dependency chains, 4x more independent work items, no workgroup
// barriers at all, and no wave ever iterates over a splat that cannot touch its pixels.
//
// Data layout in HBM (all carved from ONE caller-owned workspace, 256-byte aligned sections):
//   header         GsrHeader (64 B)                      R needed, overflow flag, stats
//   bin_count[NB], bin_count_fb[NB], bin_cursor[NB]      u32 per bin (GSR_CPAD = stride in words; with the workgroup-aggregated
//                                                        binning a bin receives only a handful of atomics, so dense counters
//                                                        are as fast as one-per-128-byte-line ones and the scan reads them coalesced)
//   bin_offset[NB+1], wg_order[NB/4], scan_part[...]     dense exclusive offsets; work-ordered workgroup list (busy first)
//   splats[P]      48-byte records {x,y,A,B | C,op,r,g | b,depth,binrect_lo,binrect_hi}: everything the compositing
//                  kernels gather per instance sits in one record (1-2 cache lines per gather instead of 3 arrays)
//   binrec[P]      16 bytes {depth bits, bin rect lo, hi, mask}: all k_scatter reads of a Gaussian.  mask bit k = the k-th cell (row-major) of its bin
//                  rect passed the exact ellipse/bin test, computed once by k_preprocess (rects of more than 32 cells carry the bits of the test's
//                  threshold instead and are re-tested by k_scatter from the splat record)
//   wg_tab[P/512][4+2048]   per binning workgroup: bin box + per-bin instance counts of its GSR_BIN_THREADS = 512 Gaussians, recorded by k_preprocess
//                  and consumed by k_scatter (which then skips the box reduction, the table clearing and the counting loop)
//   keys[cap]      u64 (depth_bits << 32 | gaussian id), binned, then sorted in LDS per bin
//   point_list[cap] u32 sorted gaussian ids (what the compositing kernels walk)
//   final_T[H*W], n_contrib[H*W]                         per-pixel state kept for the backward
//   --- backward-only tail (a forward-only caller may pass a workspace without it) ---
//   goff[P], gscan_part[P/512+1]    exclusive prefix of every Gaussian's bin-rect area (its slots in inst_valid / inst_grad), kept as
//                  (prefix inside its 512-block, prefix of the blocks): both fall out of the forward for free
//   inst_valid[cap] one byte per (Gaussian, k-th cell of its bin rect) slot: cleared by k_scatter, set by the compositing backward
//                  when it writes that instance's record
//   inst_grad[cap] 32-byte records of per-INSTANCE partial sums {dcolor rgb, dmean2D xy, dconic xx xy yy}, GAUSSIAN-MAJOR (indexed by
//                  slot); inst_dop[cap] the ninth sum (dopacity) as its own float array, so that a record is exactly one aligned
//                  32-byte sector: a scattered store of a whole sector needs no merging and no read-modify-write (36-byte packed
//                  records measured 113 MB written for 71 MB of payload).  No float atomics at all (measured 20-30 Mops/ms on
//                  MI355X, tools/ubench/), gradients bit-reproducible run to run.  The compositing backward SCATTERS its records,
//                  k_preprocess_bwd STREAMS them (a Gaussian's slots are contiguous, neighbouring threads own neighbouring slots).
//                  The other way round -- records in sorted-list order, written coalesced and gathered per Gaussian -- made every
//                  gather pull a 128-byte line for 48 useful bytes (measured: 253 MB fetched per launch against 127 MB
//                  algorithmic): scattered READS are what costs on this memory system, scattered writes cost the sectors they touch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpsgs.h"

#define GSR_TILE 16 // upstream's tile edge: defines rect membership and the reported radii semantics
#define GSR_BIN 8   // our work-item edge: one wave64 per 8x8 pixels
#define GSR_BINS_PER_WG 1 // compositing workgroup = ONE wave64 = one bin: the dispatcher then balances CUs at wave granularity
#define GSR_CPAD 1       // u32 stride of the per-bin counters / cursors (32 = one 128-byte line each: measured no faster)
#define GSR_SCAN_BLOCK 1024
#ifndef GSR_BIN_THREADS
// Gaussians per binning workgroup (k_preprocess / k_scatter) = block of the per-Gaussian slot prefix (goff / gscan_part).  1,024 until round 5: a
// 600 k-Gaussian view was 586 workgroups of 16 waves for 512 resident slots -- a second, nearly empty round of 74 workgroups on 74 CUs.  Measured
// (tools/stage_times.py, config 2 / the untrained-heads regime): k_preprocess 33.8 -> 28.3 / 53 -> 43 us, k_scatter 15.5 -> 14.8 / 106 -> 101; 256: no
// further gain for k_preprocess, k_scatter 18.8.
#define GSR_BIN_THREADS 512
#endif
#define GSR_BIN_SHIFT (GSR_BIN_THREADS == 1024 ? 10 : GSR_BIN_THREADS == 512 ? 9 : 8)
static_assert(GSR_BIN_THREADS == 1024 || GSR_BIN_THREADS == 512 || GSR_BIN_THREADS == 256, "binning workgroup size");
#define GSR_BLOCK_TAB 2048   // bins in a binning workgroup's direct-indexed LDS table

struct __attribute__((aligned(16))) GsrSplat {
    float x, y, A, B;        // pixel-space mean, conic xx, xy
    float C, op, r, g;       // conic yy, opacity, colour
    float b, depth;          // colour, view-space depth
    uint32_t bin_lo, bin_hi; // bx0 | by0<<16 , bx1 | by1<<16  (bin units, exclusive upper; empty = not listed anywhere)
};
static_assert(sizeof(GsrSplat) == 48, "splat record must be 48 bytes");

struct __attribute__((aligned(32))) GsrGradAcc {
    float dr, dg, db, dmx;    // dL/dcolor, dL/dmean2D.x (NDC-scaled)
    float dmy, cxx, cxy, cyy; // dL/dmean2D.y, dL/dconic (xy holds HALF the true off-diagonal gradient, like upstream)
};                            // dL/dopacity lives in inst_dop[] -- or, when nobody asked for dL/dcolour (GsrBwdParams::dop_in_record), in `dr`
static_assert(sizeof(GsrGradAcc) == 32, "grad record must be exactly one 32-byte sector");

// Per-bin list ranges as the kernels behind the binning see them.  Two forms (GsrViewExt.bin_capacity):
//   scanned lists (cap == 0): bin b owns [offset[b], offset[b + 1]) of keys / point_list -- the exclusive scan of the counts (k_scan_b);
//   DIRECT lists (cap > 0, round 6): bin b owns [b * cap, b * cap + count[b] + count_fb[b]) -- a fixed-capacity segment per bin, so that an instance's slot
//     follows from the base k_preprocess' count atomic returned without any offsets: the scatter pass needs no scan in front of it, and what is left
//     of the scan (totals, work order, slot prefix) leaves the critical path (it rides in the scatter launch).  A view whose longest list exceeds cap is reported
//     as an overflow (header.max_tile_count > cap) and the caller repeats it with scanned lists.
struct GsrBins {
    const uint32_t *offset, *count, *count_fb;  // (direct lists: a bin's length is count + count_fb -- the recorded workgroups' instances + the unrecorded ones')
    uint32_t cap;
};

struct GsrLayout {
    size_t header, bin_count, bin_count_fb, bin_offset, bin_cursor, wg_order, scan_part, splats, binrec, wg_tab, keys, point_list, final_T, n_contrib;
    uint32_t bin_cap;  // 0: scanned lists; else the per-bin capacity of direct lists
    int64_t key_cap;   // entries of keys / point_list: the instance capacity, or NB * bin_cap
    size_t total_fwd;  // bytes a forward-only workspace needs
    size_t goff, gscan_part, gprefix, inst_valid, inst_dop, inst_grad, total;
    int gx, gy;   // 16x16 tile grid (upstream semantics)
    int bx, by;   // bin grid: bx = ceil(W/8) rounded up to a multiple of 4, by = ceil(H/8)
    int bx_real;  // ceil(W/8)
    int NB;       // bx * by
    int NWG;      // NB / GSR_BINS_PER_WG compositing workgroups
    int NSB;      // scan blocks = ceil(max(NB, indices of the patch work order) / GSR_SCAN_BLOCK)
};

static inline size_t gsr_align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static inline GsrLayout gsr_layout(int P, int W, int H, int64_t cap, uint32_t bin_cap = 0u) {
    GsrLayout L;
    L.bin_cap = bin_cap;
    L.gx = (W + GSR_TILE - 1) / GSR_TILE;
    L.gy = (H + GSR_TILE - 1) / GSR_TILE;
    L.bx_real = (W + GSR_BIN - 1) / GSR_BIN;
    L.bx = (L.bx_real + 3) / 4 * 4;
    L.by = (H + GSR_BIN - 1) / GSR_BIN;
    L.NB = L.bx * L.by;
    L.NWG = L.NB / GSR_BINS_PER_WG;
    {
        const int nt = ((L.bx + 7) / 8) * ((L.by + 7) / 8) * 64;  // gsr_binning.hip: tiled_bin()
        L.NSB = ((nt > L.NB ? nt : L.NB) + GSR_SCAN_BLOCK - 1) / GSR_SCAN_BLOCK;
    }
    size_t o = 0;
    const size_t p = (size_t)(P > 0 ? P : 1), t = (size_t)(L.NB > 0 ? L.NB : 1), c = (size_t)(cap > 0 ? cap : 1);
    const size_t npix = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1);
    L.header = o;     o = gsr_align_up(o + sizeof(GsrHeader));   // header + scan_part + bin_count + bin_count_fb are zeroed by ONE memset
    {   // two 64-bit words per scan block (+ padding); direct lists: one per scan WAVE of the scatter launch (64 patch-order indices each)
        const size_t nt = (size_t)((L.bx + 7) / 8) * ((L.by + 7) / 8) * 64;
        const size_t nscan = ((nt > t ? nt : t) + 63) / 64;
        // direct lists: one word per scan wave of the scatter launch (whole workgroups of four)
        const size_t a = ((size_t)L.NSB + 1) * 32, b = bin_cap ? (nscan + 8) * 8 : 0;
        L.scan_part = o;  o = gsr_align_up(o + (a > b ? a : b));
    }
    L.bin_count = o;  o = gsr_align_up(o + t * 4 * GSR_CPAD);
    L.bin_count_fb = o; o = gsr_align_up(o + t * 4 * GSR_CPAD);  // instances counted by workgroups whose bins do not fit the LDS table (see gsr_block_bin)
    L.bin_cursor = o; o = gsr_align_up(o + t * 4 * GSR_CPAD);  // (inside the zeroed range: with direct lists the unrecorded workgroups' cursors start at 0)
    L.bin_offset = o; o = gsr_align_up(o + (t + 1) * 4);
    L.wg_order = o;   o = gsr_align_up(o + (t / GSR_BINS_PER_WG + 1) * 4);
    L.splats = o;     o = gsr_align_up(o + p * sizeof(GsrSplat));
    L.binrec = o;     o = gsr_align_up(o + p * 16);
    L.wg_tab = o;     o = gsr_align_up(o + ((p + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS) * (size_t)(4 + GSR_BLOCK_TAB) * 4);
    L.key_cap = bin_cap ? (int64_t)t * (int64_t)bin_cap : (int64_t)c;
    L.keys = o;       o = gsr_align_up(o + (size_t)L.key_cap * 8);
    L.point_list = o; o = gsr_align_up(o + (size_t)L.key_cap * 4);
    L.final_T = o;    o = gsr_align_up(o + npix * 4);
    L.n_contrib = o;  o = gsr_align_up(o + npix * 4);
    L.total_fwd = o;
    L.goff = o;       o = gsr_align_up(o + (p + 1) * 4);
    L.gscan_part = o; o = gsr_align_up(o + (p / GSR_BIN_THREADS + 2) * 4);
    // direct lists: the prefix is written beside the counts, not over them (the totals workgroups of the scatter launch read the counts while its scan waves scan them)
    L.gprefix = bin_cap ? o : L.gscan_part;
    if (bin_cap) o = gsr_align_up(o + (p / GSR_BIN_THREADS + 2) * 4);
    L.inst_valid = o; o = gsr_align_up(o + c);
    L.inst_dop = o;   o = gsr_align_up(o + c * 4);
    L.inst_grad = o;  o = gsr_align_up(o + c * sizeof(GsrGradAcc));
    L.total = o;
    return L;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void gsr_bin_range(const GsrBins &b, uint32_t bin, uint32_t &r0, uint32_t &r1) {
    if (b.cap) {  // (kernel-argument uniform)
        const uint32_t n = b.count[bin] + b.count_fb[bin];
        r0 = bin * b.cap;
        r1 = r0 + (n < b.cap ? n : b.cap);  // (a longer list is an overflow: nothing behind the scan runs; the clamp only keeps a debug read inside its segment)
    } else {
        r0 = b.offset[bin];
        r1 = b.offset[bin + 1];
    }
}

// Workgroup-aggregated binning.  Same-address device atomics serialise at ~0.1-0.2 us each on MI355X (measured,
// tools/ubench/atomic_bench*.hip), and a body bin receives ~450 instances, so per-instance -- or even per-wave --
// atomics on the per-bin counters cost >100 us.  Pixel-Gaussians arrive in source-raster order, so the bins touched
// by one 1024-thread workgroup form a small rectangle of the bin grid: the workgroup histograms its instances into a
// direct-indexed LDS table over that rectangle (ds_add, which also hands every instance its rank) and then issues ONE
// global atomic per touched bin (reserve(bin, count) -> base).  emit(pos, cell, bin) is called once per instance with its slot,
// the row-major index of the bin inside the Gaussian's own bin rect, and the bin.
// Incoherent input (bounding rectangle > GSR_BLOCK_TAB bins) falls back to one global atomic per instance.

// Exact (Gaussian, bin) culling inside the bin rect.  alpha = op*exp(-q/2) >= 1/255  <=>  q(d) = A dx^2 + 2B dx dy + C dy^2
// <= 2 ln(255 op).  The bin is listed only if the MINIMUM of q over the rectangle of its pixel centres is below that
// (slightly inflated) threshold: q is convex, so the minimum is 0 if the centre is inside and otherwise lies on one of the
// (at most two) edges facing the centre, where it is a clamped 1-D parabola minimum.  Drops the ~18 % of bounding-box
// instances whose ellipse misses the bin's corner.  Evaluated by BOTH the counting and the scatter kernel: it must be
// bit-reproducible, hence no FMA contraction in here (the two kernels live in translation units with different flags).
struct GsrHit {
    float x, y, A, B, C, thr, rA, rC;
};
// 1 / x as ONE instruction (v_rcp_f32, 1 ulp) where the value only places the minimiser of a parabola along a bin edge: q is flat there, so an ulp of
// the position is an ulp SQUARED of q, against a threshold inflated by 0.2 %.  The correctly rounded division (this library's compile flag) is ~12
// instructions, twice per Gaussian in each of the VALU-bound kernels k_preprocess and k_scatter.  Deterministic: both passes execute the same instruction
// on the same stored record.
__device__ __forceinline__ float gsr_rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
// tau = logf(255 op), which the caller has computed already (a second logf() per Gaussian was ~25 vector instructions of a VALU-bound kernel)
__device__ __forceinline__ GsrHit gsr_hit_setup(float x, float y, float A, float B, float C, float tau) {
    _Pragma("clang fp contract(off)")
    GsrHit h;
    h.x = x; h.y = y; h.A = A; h.B = B; h.C = C;
    h.thr = 2.f * tau * 1.002f + 0.02f;
    h.rA = gsr_rcp_approx(A);
    h.rC = gsr_rcp_approx(C);
    return h;
}
// The same test from a threshold that was computed elsewhere: k_scatter rebuilds the predicate of a Gaussian whose rect is too large for
// the cached mask from the splat record plus the threshold k_preprocess STORED in the mask word -- it must not call logf() itself: the
// math library's polynomial is contracted differently in translation units with different flags (measured: one v_add became a v_fmac,
// thresholds one ulp apart, one (Gaussian, bin) decision in ~10^7 flipped between the count and the scatter pass -- a key slot was then
// never written and the compositing kernel read a random Gaussian id: config 3 with random network weights, splats of ~40 px radius).
__device__ __forceinline__ GsrHit gsr_hit_from_threshold(float x, float y, float A, float B, float C, float thr) {
    _Pragma("clang fp contract(off)")
    GsrHit h;
    h.x = x; h.y = y; h.A = A; h.B = B; h.C = C;
    h.thr = thr;
    h.rA = gsr_rcp_approx(A);
    h.rC = gsr_rcp_approx(C);
    return h;
}
// minimum of the quadratic form over the rectangle of pixel centres [X0, X1] x [Y0, Y1] against the (inflated) threshold.
// The fixed inflation of the threshold (0.2 % + 0.02) covers the difference between this evaluation order and the compositing
// kernels' exp(); what it cannot cover is the rounding of q itself when the conic is strongly anisotropic or correlated and the terms
// A dx^2, 2 B dx dy, C dy^2 are far larger than their sum: every candidate is therefore credited 1e-5 of the sum of the terms'
// MAGNITUDES (fp32 evaluates q to a few 1e-7 of that sum), so a pair the compositor could still blend is never dropped.
__device__ __forceinline__ float gsr_q_lower(const GsrHit &h, float dx, float dy) {
    _Pragma("clang fp contract(off)")
    const float a = h.A * dx * dx, b = 2.f * h.B * dx * dy, c = h.C * dy * dy;
    return (a + b + c) - 1e-5f * (fabsf(a) + fabsf(b) + fabsf(c));
}
__device__ __forceinline__ bool gsr_rect_hit(const GsrHit &h, float X0, float X1, float Y0, float Y1) {
    _Pragma("clang fp contract(off)")
    const float cx = fminf(fmaxf(h.x, X0), X1), cy = fminf(fmaxf(h.y, Y0), Y1);
    if (cx == h.x && cy == h.y) return true;  // centre inside the rectangle
    float best = 3.0e38f;
    if (cx != h.x) {  // vertical edge x = cx faces the centre
        const float dx = cx - h.x;
        const float yy = fminf(fmaxf(h.y - h.B * dx * h.rC, Y0), Y1);
        best = fminf(best, gsr_q_lower(h, dx, yy - h.y));
    }
    if (cy != h.y) {  // horizontal edge y = cy faces the centre
        const float dy = cy - h.y;
        const float xx = fminf(fmaxf(h.x - h.B * dy * h.rA, X0), X1);
        best = fminf(best, gsr_q_lower(h, xx - h.x, dy));
    }
    return best <= h.thr;
}
__device__ __forceinline__ bool gsr_bin_hit(const GsrHit &h, int bxi, int byi) {
    const float X0 = (float)(bxi * GSR_BIN), Y0 = (float)(byi * GSR_BIN);
    return gsr_rect_hit(h, X0, X0 + (float)(GSR_BIN - 1), Y0, Y0 + (float)(GSR_BIN - 1));
}

// ---- LARGE rects (more than 32 cells): one x-interval per bin row instead of a test per cell ------------------------------------------
// With freshly initialised networks (BASELINE configs 3 / 4 offline, and the first iterations of stage-2 training) every scale sits at its 0.01 m
// clamp: ~100-cell rects, ~55 hits per Gaussian, 3 x 10^7 instances per 2048^2 view, and the per-cell test above (~45 VALU instructions, evaluated
// by the count AND the scatter pass) made k_preprocess / k_scatter 160 / 215 us there (28 / 16 at config 2).  {q <= thr} is convex, so inside
// the band of pixel rows [Y0, Y0 + 7] of one bin row its x-projection is ONE interval [xl, xr]: the extreme points are those of the ellipse
// (at dy = -+kk) clamped into the band, where x = x0 - (B/A) dy +- sqrt(A thr - det dy^2) / A.  A cell [X0, X0 + 7] is listed iff it overlaps
// [xl - 1/4, xr + 1/4]: two square roots per ROW, then a count per cell.  This is NOT the same predicate bit for bit as gsr_rect_hit (it lists
// a superset: the quarter pixel, and thr is already inflated by 0.2 % + 0.02, four orders of magnitude above the rounding of the discriminant);
// extra pairs are harmless (the compositor skips them with alpha < 1/255), dropped ones would not be.  What matters is that the COUNT pass
// (k_preprocess) and the SCATTER pass (k_scatter) take identical decisions: both evaluate this very function on the stored record + the stored
// threshold, with IEEE +,-,*,/,sqrt,min,max only (both translation units: -ffp-contract=off, correctly rounded divide / sqrt; no libm call).
// Ill-conditioned conics (det < 1 % of A C: needle-shaped splats, where det itself cancels) keep the per-cell test.
struct GsrRowSpan {
    float x, y, BA, rA, AT, det, kk;
    int ok;
};
__device__ __forceinline__ GsrRowSpan gsr_rows_setup(const GsrHit &h) {
    _Pragma("clang fp contract(off)")
    GsrRowSpan r;
    r.x = h.x; r.y = h.y; r.rA = h.rA;
    r.BA = h.B * h.rA;
    r.AT = h.A * h.thr;
    const float ac = h.A * h.C;
    r.det = ac - h.B * h.B;
    r.ok = (h.A > 0.f && h.C > 0.f && h.thr > 0.f && r.det > 0.f && r.det >= 0.01f * ac) ? 1 : 0;  // (any NaN compares false)
    r.kk = r.ok ? h.B * sqrtf(h.thr / (h.C * r.det)) : 0.f;  // the ellipse's rightmost point lies at dy = -kk, its leftmost at dy = +kk
    return r;
}
// cells [xa, xb) of bin row byi, clipped to the rect's columns [x0, x1); xa == xb: none
__device__ __forceinline__ void gsr_row_cells(const GsrRowSpan &r, int byi, int x0, int x1, int &xa, int &xb) {
    _Pragma("clang fp contract(off)")
    const float d0 = (float)(byi * GSR_BIN) - r.y, d1 = d0 + (float)(GSR_BIN - 1);
    const float dyR = fminf(fmaxf(-r.kk, d0), d1), dyL = fminf(fmaxf(r.kk, d0), d1);
    const float discR = r.AT - r.det * dyR * dyR, discL = r.AT - r.det * dyL * dyL;
    const float xr = (r.x - r.BA * dyR) + sqrtf(fmaxf(discR, 0.f)) * r.rA + 0.25f;
    const float xl = (r.x - r.BA * dyL) - sqrtf(fmaxf(discL, 0.f)) * r.rA - 0.25f;
    // the band misses the (inflated) ellipse iff both discriminants are negative; 8 c + 7 >= xl and 8 c <= xr otherwise
    const bool some = discR >= 0.f || discL >= 0.f;
    const int lo = (int)ceilf((xl - (float)(GSR_BIN - 1)) * 0.125f), hi = (int)floorf(xr * 0.125f) + 1;
    xa = some ? max(lo, x0) : x0;
    xb = some ? min(hi, x1) : x0;
    if (xb < xa) xb = xa;
}

// Bit k of a hit mask = cell k (row-major inside the rect [x0,x1) x [y0,y1)) passed the exact test.  Rects of more than 32
// cells are not cached: their mask word carries the bits of the test's threshold instead (GsrHit::thr) and their cells are re-tested,
// with that very threshold, wherever the mask is consumed.
// hit predicate backed by a cached mask (falls back to the exact test for uncached rects)
struct GsrMaskedHit {
    GsrHit h;
    GsrRowSpan rs;
    uint32_t mask;
    int x0, y0, w;
    bool big, rows;
    __device__ __forceinline__ void span(int y, int &xa, int &xb) const {  // the columns of row y worth looking at
        if (rows) gsr_row_cells(rs, y, xa, xb, xa, xb);
    }
    __device__ __forceinline__ bool operator()(int x, int y) const {
        if (rows) return true;  // span() already was the test
        if (big) return gsr_bin_hit(h, x, y);
        return (mask >> ((y - y0) * w + (x - x0))) & 1u;
    }
};
__device__ __forceinline__ GsrMaskedHit gsr_masked_hit(const GsrHit &h, uint32_t mask, uint32_t lo, uint32_t hi) {
    GsrMaskedHit m;
    m.h = h; m.mask = mask;
    m.x0 = lo & 0xffff; m.y0 = lo >> 16; m.w = (int)(hi & 0xffff) - m.x0;
    const int area = m.w * ((int)(hi >> 16) - m.y0);
    m.big = area > 32;
    m.rs.ok = 0;
    if (m.big) m.rs = gsr_rows_setup(h);  // (a division and a square root nobody needs for the ~4-cell rects of trained scales: k_preprocess skips them the same way)
    m.rows = m.big && m.rs.ok;
    return m;
}

// ---- the two binning passes of a workgroup of (up to) GSR_BIN_THREADS consecutive Gaussians ----------------------------------------------------------
// COUNT pass (k_preprocess, gsr_block_bin<false>): histogram the workgroup's instances into the LDS table over its bin box, then ONE returning
// global atomic per touched bin: bin_count[bin] += count.  The value it returns is this workgroup's BASE inside the bin's list; it is recorded
// in `tab` = {bx0, by0, bw, bh, base + 1 per table entry (0: no instance)} for the scatter pass of the same Gaussians.
// SCATTER pass (k_scatter, gsr_block_emit_*): slot = bin_offset[bin] + recorded base + rank inside the workgroup (LDS atomic) -- no global atomic,
// no counting loop, no box reduction (round 5; until then the scatter pass reserved its slots with a second returning atomic per bin).
// INCOHERENT input -- a bin box that does not fit the table (normal case: the ONE workgroup whose Gaussians straddle the end of the first
// source view and the start of the second: a box as tall as the subject) -- is counted band by band into a SECOND counter array
// (reserve_fb -> bin_count_fb) and not recorded (tab[2] = -1); the scan places those instances BEHIND the recorded ones of their bin
// (bin_cursor = offset + recorded count) and the scatter pass of that workgroup rebuilds its tables and reserves from that cursor
// (gsr_block_bin<true>).  Both passes must take identical (Gaussian, bin) decisions: `hit`.
#define GSR_WG_TAB_WORDS (4 + GSR_BLOCK_TAB)
// NT = threads of the workgroup (one Gaussian per thread and call); s_cnt / s_base: GSR_BLOCK_TAB words of LDS each (s_base: EMIT only), s_box: 4.
template <bool EMIT, int NT, typename Hit, typename Reserve, typename ReserveFb, typename Emit>
__device__ __forceinline__ void gsr_block_bin(uint32_t *s_cnt, uint32_t *s_base, int *s_box, uint32_t lo, uint32_t hi, int bx, Hit hit, Reserve reserve,
                                              ReserveFb reserve_fb, Emit emit, uint32_t *tab = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff, y1 = hi >> 16;
    const bool has = (x1 > x0) && (y1 > y0);
    __syncthreads();  // (a previous call's tables are done with)
    if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -1; s_box[3] = -1; }
    // the wave's bin box: {min x, min y} and {max x, max y} as two PACKED pairs of 16-bit numbers (bin coordinates stay below 8,192): 12 lane exchanges and 12
    // v_pk_min / v_pk_max_i16 instead of 24 + 24 -- ~4 % of a VALU-bound kernel's instructions (round 6)
    typedef short gsr_s16x2 __attribute__((ext_vector_type(2)));
    gsr_s16x2 mn = {(short)(has ? x0 : 0x7fff), (short)(has ? y0 : 0x7fff)}, mx = {(short)(has ? x1 : -1), (short)(has ? y1 : -1)};
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const gsr_s16x2 on = __builtin_bit_cast(gsr_s16x2, __shfl_xor(__builtin_bit_cast(int, mn), d, 64));
        const gsr_s16x2 ox = __builtin_bit_cast(gsr_s16x2, __shfl_xor(__builtin_bit_cast(int, mx), d, 64));
        mn = __builtin_elementwise_min(mn, on);
        mx = __builtin_elementwise_max(mx, ox);
    }
    const int mnx = mn.x, mny = mn.y, mxx = mx.x, mxy = mx.y;
    __syncthreads();
    if (lane == 0 && mxx >= 0) {
        atomicMin(&s_box[0], mnx); atomicMin(&s_box[1], mny); atomicMax(&s_box[2], mxx); atomicMax(&s_box[3], mxy);
    }
    __syncthreads();
    const int bx0 = s_box[0], by0 = s_box[1], bw = s_box[2] - bx0, bh = s_box[3] - by0;
    if (s_box[2] < 0) {  // nothing listed in this workgroup (uniform)
        if (tab && tid < 4) tab[tid] = 0u;
        return;
    }
    const int area = bw * bh;
    if (tab && tid < 4) tab[tid] = area > GSR_BLOCK_TAB ? 0xffffffffu : (uint32_t)(tid == 0 ? bx0 : tid == 1 ? by0 : tid == 2 ? bw : bh);
    if (area > GSR_BLOCK_TAB && bw <= GSR_BLOCK_TAB) {
        // The workgroup's bins do not fit the table (uniform branch): the table is applied to BANDS of bin rows, one after the other.  (Until round 4
        // such a workgroup fell back to one GLOBAL atomic per instance: invisible with ~5 instances per Gaussian, but with scales at their clamp it
        // is 70 dependent global atomics per thread in one workgroup -- ~100 us during which the rest of the chip had long finished.)
        const int band_h = GSR_BLOCK_TAB / bw;  // >= 1 bin rows per pass
        for (int yb = by0; yb < by0 + bh; yb += band_h) {
            const int yb1 = min(yb + band_h, by0 + bh), barea = (yb1 - yb) * bw;
            const int ya = max(y0, yb), yz = min(y1, yb1);
            __syncthreads();  // the previous band's table is done with
            for (int t = tid; t < barea; t += NT) s_cnt[t] = 0u;
            __syncthreads();
            if (has)
                for (int y = ya; y < yz; y++) {
                    int xa = x0, xb = x1;
                    hit.span(y, xa, xb);
                    for (int x = xa; x < xb; x++)
                        if (hit(x, y)) atomicAdd(&s_cnt[(y - yb) * bw + (x - bx0)], 1u);
                }
            __syncthreads();
            for (int t = tid; t < barea; t += NT) {
                const uint32_t c = s_cnt[t];
                if (c) {
                    const int ty = t / bw, tx = t - ty * bw;
                    const uint32_t base = reserve_fb((yb + ty) * bx + bx0 + tx, c);
                    if (EMIT) { s_base[t] = base; s_cnt[t] = 0u; }
                }
            }
            if (EMIT) {
                __syncthreads();
                if (has)
                    for (int y = ya; y < yz; y++) {
                        int xa = x0, xb = x1;
                        hit.span(y, xa, xb);
                        for (int x = xa; x < xb; x++) {
                            if (!hit(x, y)) continue;
                            const int t = (y - yb) * bw + (x - bx0);
                            emit(s_base[t] + atomicAdd(&s_cnt[t], 1u), (uint32_t)((y - y0) * (x1 - x0) + (x - x0)), (uint32_t)(y * bx + x));
                        }
                    }
            }
        }
        return;
    }
    if (area > GSR_BLOCK_TAB) {  // a box wider than the whole table (images beyond 16,384 pixels across): plain per-instance atomics (uniform branch)
        if (has)
            for (int y = y0; y < y1; y++) {
                int xa = x0, xb = x1;
                hit.span(y, xa, xb);
                for (int x = xa; x < xb; x++) {
                    if (!hit(x, y)) continue;
                    const uint32_t pos = reserve_fb(y * bx + x, 1u);
                    if (EMIT) emit(pos, (uint32_t)((y - y0) * (x1 - x0) + (x - x0)), (uint32_t)(y * bx + x));
                }
            }
        return;
    }
    for (int t = tid; t < area; t += NT) s_cnt[t] = 0u;
    __syncthreads();
    if (has)
        for (int y = y0; y < y1; y++) {
            int xa = x0, xb = x1;
            hit.span(y, xa, xb);
            for (int x = xa; x < xb; x++)
                if (hit(x, y)) atomicAdd(&s_cnt[(y - by0) * bw + (x - bx0)], 1u);
        }
    __syncthreads();
    for (int t = tid; t < area; t += NT) {
        const uint32_t c = s_cnt[t];
        uint32_t base = 0u;
        if (c) {
            const int ty = t / bw, tx = t - ty * bw;
            base = reserve((by0 + ty) * bx + bx0 + tx, c);
            if (EMIT) { s_base[t] = base; s_cnt[t] = 0u; }
        }
        if (tab) tab[4 + t] = c ? base + 1u : 0u;  // this workgroup's base inside the bin's list, for the scatter pass
    }
    if (!EMIT) return;
    __syncthreads();
    if (has)
        for (int y = y0; y < y1; y++) {
            int xa = x0, xb = x1;
            hit.span(y, xa, xb);
            for (int x = xa; x < xb; x++) {
                if (!hit(x, y)) continue;
                const int t = (y - by0) * bw + (x - bx0);
                emit(s_base[t] + atomicAdd(&s_cnt[t], 1u), (uint32_t)((y - y0) * (x1 - x0) + (x - x0)), (uint32_t)(y * bx + x));
            }
        }
}
// Scatter pass from the recorded table (k_scatter sets the tables up itself): per Gaussian, rank inside the workgroup by LDS atomic, then emit.
template <typename Hit, typename Emit>
__device__ __forceinline__ void gsr_block_emit_one(uint32_t *e_cnt, const uint32_t *e_base, int bx0, int by0, int bw, int bx, uint32_t lo, uint32_t hi, Hit hit, Emit emit) {
    const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff, y1 = hi >> 16;
    if ((x1 > x0) && (y1 > y0))
        for (int y = y0; y < y1; y++) {
            int xa = x0, xb = x1;
            hit.span(y, xa, xb);
            for (int x = xa; x < xb; x++) {
                if (!hit(x, y)) continue;
                const int t = (y - by0) * bw + (x - bx0);
                emit(e_base[t] + atomicAdd(&e_cnt[t], 1u), (uint32_t)((y - y0) * (x1 - x0) + (x - x0)), (uint32_t)(y * bx + x));
            }
        }
}
#endif

// Launchers implemented in the .hip files (host side).  All enqueue on `s` and never synchronise.
struct GsrFwdParams {
    int P, W, H, gx, gy, bx, by, bx_real;
    const float *means3D, *colors, *opacities, *scales, *rotations;
    float scale_modifier, tanfovx, tanfovy;
    const float *view, *proj, *bg;
    float *out_color;
    int *radii;
    int64_t cap;
    uint32_t *goff, *gpart;  // backward tail of the workspace (NULL for a forward-only workspace)
    const uint32_t *row_range;  // device {begin, end} or NULL: the view's Gaussians are rows [begin, end) of the batch-wide arrays, P = capacity
    // the other half of the upstream interface (GsrViewExt): SH colours instead of `colors`, precomputed covariances instead of scales + rotations
    const float *shs, *campos, *cov3D_precomp;  // [rows, sh_coeffs, 3], [3], [rows, 6]; NULL = not used
    uint32_t sh_degree, sh_coeffs;
    float fx, fy;  // focal lengths in pixels, W / (2 tanfovx): set by gsr_launch_preprocess (two IEEE divisions per THREAD of a VALU-bound kernel otherwise)
};

void gsr_launch_preprocess(const GsrFwdParams &p, GsrSplat *splats, uint4 *binrec, uint32_t *wg_tab, uint32_t *bin_count, uint32_t *bin_count_fb, GsrHeader *hdr,
                           hipStream_t s);
void gsr_launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);
void gsr_launch_scan(const uint32_t *bin_count, const uint32_t *bin_count_fb, uint32_t *bin_offset, uint32_t *bin_cursor, uint32_t *wg_order, uint4 *scan_part, int NB, int bx,
                     int by, int64_t cap, GsrHeader *hdr, uint32_t *gpart, int n_gblocks, uint32_t *host_hdr, uint32_t host_seq, bool no_large_sort, uint32_t order_hint,
                     hipStream_t s);
void gsr_launch_scatter(int P, const uint32_t *row_range, int bx, const GsrSplat *splats, const uint4 *binrec, const uint32_t *wg_tab, const uint32_t *bin_offset,
                        uint32_t *bin_cursor, uint64_t *keys, const GsrHeader *hdr, hipStream_t s);
// direct lists: the same scatter pass with slot = bin * bin_cap + recorded base + rank (no offsets to gather).  Workgroups in front of the scattering ones
// do what is left of the scan from the counters k_preprocess left: the totals -- R, longest list, gradient slots -- and the header, also to the host (early
// capacity notification); the work order of the compositing waves; the slot prefix of the gradient records (gpart: counts in, gprefix: prefix out)
void gsr_launch_scatter_direct(int P, const uint32_t *row_range, int bx, int by, int NB, const GsrSplat *splats, const uint4 *binrec, const uint32_t *wg_tab, const uint32_t *bin_count,
                               const uint32_t *bin_count_fb, uint32_t *bin_cursor, uint32_t bin_cap, uint64_t *keys, GsrHeader *hdr, int64_t slot_cap, const uint32_t *gpart,
                               uint32_t *gprefix, int n_gblocks, uint32_t *wg_order, uint4 *scan_part, uint32_t order_hint, uint32_t *host_hdr, uint32_t host_seq, hipStream_t s);
void gsr_launch_sort(int NB, const uint32_t *bin_offset, const uint32_t *wg_order, uint64_t *keys, uint32_t *point_list,
                     const GsrHeader *hdr, bool no_large_sort, hipStream_t s);
// lists of <= 1,024 keys, either list form (direct lists hold nothing longer): one wave per busy bin
void gsr_launch_sort_direct(int NB, GsrBins bins, const uint32_t *wg_order, uint64_t *keys, uint32_t *point_list, const GsrHeader *hdr, hipStream_t s);
#define GSR_DIRECT_MAX_BINS 65536  // direct lists: the scan waves of the scatter launch poll each other's partials (all resident: <= 1,024 of them)
#define GSR_DIRECT_MAX_CAP 1024    // ... and a bin's list is sorted by ONE wave (k_sort_wave's classes)
void gsr_launch_composite_fwd(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                              const uint32_t *point_list, const float *bg, float *out_color, float *final_T, uint32_t *n_contrib, const GsrHeader *hdr,
                              uint8_t *inst_valid /* training workspace: the record flags are cleared here; NULL otherwise */, hipStream_t s);
void gsr_launch_composite_bwd(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                              const uint32_t *point_list, const float *bg, const float *dL_dpix, const float *final_T, const uint32_t *n_contrib,
                              const uint32_t *goff, const uint32_t *gpart, uint8_t *inst_valid, float *inst_dop, GsrGradAcc *inst_grad, const GsrHeader *hdr,
                              hipStream_t s);
// development knob: GPSGS_DEBUG_LDS_PAD=<bytes> of unused dynamic LDS per compositing workgroup (caps the waves resident per CU, to
// measure how the kernels scale with occupancy); 0 / unset in normal use
#include <stdlib.h>
static inline unsigned gsr_debug_lds_pad() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("GPSGS_DEBUG_LDS_PAD"); v = e ? atoi(e) : 0; if (v < 0) v = 0; }
    return (unsigned)v;
}
// exponents from bf16 matrix-core tiles (gsr_composite_tiles.hip): same arguments, same results within rounding
void gsr_launch_composite_fwd_tiles(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                                    uint32_t *point_list, const float *bg, float *out_color, float *final_T, uint32_t *n_contrib, const GsrHeader *hdr,
                                    uint8_t *inst_valid /* training workspace: the record flags are cleared here; NULL otherwise */,
                                    bool keep_state /* false: inference workspace, final_T / n_contrib are not produced */,
                                    bool wave_prio /* GSR_FLAG_WAVE_PRIORITY */,
                                    const uint64_t *unsorted_keys /* direct lists: every wave sorts its own bin's list first (keys -> point_list); NULL: point_list is sorted */,
                                    hipStream_t s);
int gsr_set_wg_trace(unsigned long long *rows_device);  // development aid: per-workgroup timeline of the tile compositing kernels (NULL = off)
void gsr_launch_composite_bwd_tiles(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                                    const uint32_t *point_list, const float *bg, const float *dL_dpix, const float *final_T, const uint32_t *n_contrib,
                                    const uint32_t *goff, const uint32_t *gpart, uint8_t *inst_valid, float *inst_dop, GsrGradAcc *inst_grad, const GsrHeader *hdr,
                                    bool color_grad /* false: GSR_FLAG_NO_COLOR_GRAD, the colour sums are left out (zeros in the records) */,
                                    bool wave_prio /* GSR_FLAG_WAVE_PRIORITY */, hipStream_t s);
void gsr_launch_selftest(float *out4, hipStream_t s);
struct GsrBwdParams {
    int P, W, H;
    const float *means3D, *scales, *rotations;
    float scale_modifier, tanfovx, tanfovy;
    const float *view, *proj;
    const int *radii;
    float *dL_dmeans3D, *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dscales, *dL_drotations;
    const uint32_t *row_range;  // as in GsrFwdParams
    const float *shs, *campos, *cov3D_precomp;  // as in GsrFwdParams
    uint32_t sh_degree, sh_coeffs;
    float *dL_dsh, *dL_dcov3D;  // [rows, sh_coeffs, 3], [rows, 6]: written when the matching input is given
    int dop_in_record;          // the records were written without colour sums (tile family + GSR_FLAG_NO_COLOR_GRAD): dL/dopacity is their first float, inst_dop is not read
    float fx, fy;               // as in GsrFwdParams: set by gsr_launch_preprocess_bwd
};

#if defined(__HIPCC__)
// The rows of one view: without a row range rows [0, P); with one, rows [begin, end) of the batch-wide arrays, at most `cap` of them
// (wave-uniform scalar loads).  first = row of the view's Gaussian 0, n = number of Gaussians the kernels may touch.
__device__ __forceinline__ void gsr_view_rows(const uint32_t *__restrict__ row_range, int cap, uint32_t &first, int &n) {
    first = 0u;
    n = cap;
    if (row_range) {
        first = row_range[0];
        const uint32_t m = row_range[1] - first;
        n = m > (uint32_t)cap ? cap : (int)m;
    }
}
#endif
void gsr_launch_preprocess_bwd(const GsrBwdParams &p, const GsrSplat *splats, const uint32_t *goff, const uint32_t *gpart,
                               const uint8_t *inst_valid, const float *inst_dop, const GsrGradAcc *inst_grad, const GsrHeader *hdr, hipStream_t s);
