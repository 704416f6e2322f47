// gsr_composite_common.h -- helpers shared by the compositing kernels (gsr_composite.hip: VALU-only design of round 1,
// gsr_composite_mfma.hip: the matrix-core design): bin geometry of a wave, XCD-aware work-list mapping, same-wave LDS hand-off.
#pragma once
#include "gsr_common.h"

namespace {


constexpr int WAVE = 64;
constexpr int WAVES = GSR_BINS_PER_WG;  // 4

struct WaveGeom {
    int bin, px, py, lane, wid;
    bool inside;
    uint32_t r0, r1;
};

// Position in the work-ordered list (busy bins first, row-major) this workgroup takes.  Workgroups are handed to the 8 XCDs
// round-robin (workgroup w -> XCD w % 8) and every XCD has its own L2, while a Gaussian is listed in ~3 NEIGHBOURING bins:
// taking the list in dispatch order would put neighbours on different XCDs and every XCD would fetch its own copy of the
// shared splat records from HBM (measured: 2.2x the fetch traffic).  Instead the busy list is cut into runs of 64 consecutive
// bins (half an image row at 1024^2) that are dealt to the XCDs in turn: neighbours along a row share one L2, and every XCD
// still gets runs from all over the image (one contiguous eighth per XCD cut the traffic further but left the XCDs unevenly
// loaded: +3 % time).  The map is a bijection on [0, 512 * ceil(busy / 512)); idle bins behind it keep their place.
// (If that range does not fit the grid -- tiny images -- keep the identity.)
__device__ __forceinline__ uint32_t xcd_list_pos(uint32_t w, uint32_t busy) {
    const uint32_t span = ((busy + 511u) >> 9) << 9;
    if (span > gridDim.x || w >= span) return w;
    const uint32_t x = w & 7u, q = w >> 3;
    return (((q >> 6) << 3) + x) * 64u + (q & 63u);
}

__device__ __forceinline__ WaveGeom wave_geom(int W, int H, int bx, const uint32_t *__restrict__ bin_offset,
                                              const uint32_t *__restrict__ wg_order, uint32_t list_pos) {
    WaveGeom g;
    const int tid = threadIdx.x;
    g.lane = tid & 63;
    g.wid = tid >> 6;
    const int wg = (int)wg_order[list_pos];  // work-ordered dispatch: workgroups with non-empty lists come first
    const int wgs_per_row = bx / WAVES;
    const int by_i = wg / wgs_per_row, bx_i = (wg - by_i * wgs_per_row) * WAVES + g.wid;
    g.bin = by_i * bx + bx_i;
    g.px = bx_i * GSR_BIN + (g.lane & 7);
    g.py = by_i * GSR_BIN + (g.lane >> 3);
    g.inside = g.px < W && g.py < H;
    g.r0 = bin_offset[g.bin];
    g.r1 = bin_offset[g.bin + 1];
    return g;
}

// overflowed forward: this wave's pixels become a defined blank (zero colour, T = 1, no contributor)
__device__ __forceinline__ void fwd_write_blank(const WaveGeom &g, int W, int H, float *__restrict__ out_color, float *__restrict__ final_T,
                                                uint32_t *__restrict__ n_contrib) {
    if (!g.inside) return;
    const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
    out_color[q] = 0.f; out_color[npix + q] = 0.f; out_color[2 * npix + q] = 0.f;
    final_T[q] = 1.f;
    n_contrib[q] = 0u;
}

// The exponent of a (pixel, splat) pair in log2 units, from a conic that was pre-scaled when the splat was staged
// (A2 = -0.5 log2(e) A, B2 = -log2(e) B, C2 = -0.5 log2(e) C): 6 instructions instead of 8 and v_exp_f32 directly.  Forward and backward
// MUST evaluate this same sequence (explicit fma: no contraction choices left to the compiler) so that both take the same
// alpha >= 1/255 decisions.
#define GSR_LOG2E 1.44269504088896340736f
__device__ __forceinline__ float gsr_power2(float A2, float B2, float C2, float dx, float dy) {
    return __builtin_fmaf(B2 * dx, dy, __builtin_fmaf(C2 * dy, dy, (A2 * dx) * dx));
}

__device__ __forceinline__ void wave_sync_lds() {
    // same-wave LDS hand-off: the DS queue is in order per wave, this only pins the compiler's ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


}  // namespace
