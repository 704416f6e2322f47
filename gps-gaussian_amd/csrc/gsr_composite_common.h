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

// Position in the work-ordered list (busy bins first, in patches of 8x8 bins: gsr_binning.hip tiled_bin) this workgroup takes.
// Workgroups are handed to the 8 XCDs round-robin (workgroup w -> XCD w % 8) and every XCD has its own L2, while a Gaussian is listed
// in ~3 NEIGHBOURING bins: taking the list in dispatch order would put neighbours on different XCDs and every XCD would fetch its own
// copy of the shared splat records (measured: 2.2x the fetch traffic).  Instead the busy list is cut into runs of 64 consecutive
// entries (about one 64x64-pixel patch) that are dealt to the XCDs in turn: the neighbours of a bin in BOTH directions share one L2,
// and every XCD still gets runs from all over the image (one contiguous eighth per XCD cut the traffic further but left the XCDs
// unevenly loaded: +3 % time).  The map is a bijection on [0, 512 * ceil(busy / 512)); idle bins behind it keep their place.
// (If that range does not fit the grid -- tiny images -- keep the identity.)
__device__ __forceinline__ uint32_t xcd_list_pos(uint32_t w, uint32_t busy) {
    const uint32_t span = ((busy + 511u) >> 9) << 9;
    if (span > gridDim.x || w >= span) return w;
    const uint32_t x = w & 7u, q = w >> 3;
    return (((q >> 6) << 3) + x) * 64u + (q & 63u);
}

__device__ __forceinline__ WaveGeom wave_geom(int W, int H, int bx, const GsrBins &bins,
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
    gsr_bin_range(bins, (uint32_t)g.bin, g.r0, g.r1);
    return g;
}

// Training workspaces: "no gradient record yet" for every slot of the view, written by the FORWARD compositing launch -- every workgroup clears its
// share of the flag bytes [0, num_slots) with 16-byte stores (one per lane at config 2: 2.3 MB over 16,384 workgroups).  Until round 6 k_scatter cleared
// the slots of its own Gaussians; direct lists have no scatter pass, and the slot total is only known once the scan has run.  The section is 256-byte
// aligned and padded, so whole 16-byte words up to the rounded-up slot count stay inside it (num_slots <= capacity, or the view overflowed and
// nothing runs).
__device__ __forceinline__ void clear_record_flags(uint8_t *__restrict__ inst_valid, const GsrHeader *__restrict__ hdr, int tid, int nthreads) {
    if (!inst_valid) return;
    const uint32_t n16 = (hdr->num_slots + 15u) >> 4;
    const uint32_t per = (n16 + gridDim.x - 1u) / gridDim.x;
    const uint32_t w0 = blockIdx.x * per;
    for (uint32_t k = (uint32_t)tid; k < per; k += (uint32_t)nthreads) {
        const uint32_t w = w0 + k;
        if (w < n16) reinterpret_cast<uint4 *>(inst_valid)[w] = make_uint4(0u, 0u, 0u, 0u);
    }
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


// ---- nine simultaneous wave64 sums as a butterfly REDUCE-SCATTER ------------------------------------------------------
// Summing 9 values over 64 lanes one by one costs 9 x 6 = 54 DPP adds.  A reduce-scatter halves the number of live
// registers at every level instead:  xor-32 level: v_permlane32_swap + add folds TWO values into one register (lower
// half-wave = value a, upper = value b);  xor-16 level: v_permlane16_swap + add folds two of those (even rows / odd rows);
// xor-8 level: one select pair + row_ror:8 add;  then 3 DPP adds inside each 8-lane group.  8 values: 8+4+3+3 = 18
// instructions; the 9th is only row-reduced (4 DPP adds) and its 4 row sums are added at flush time.
// Result: value k of v[0..7] sits (fully summed) in every lane of one 8-lane group, see acc_slot(); v[8]'s row sums
// sit in lanes 15, 31, 47, 63.
template <int CTRL>
__device__ __forceinline__ float dpp_add_row(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float swap_add32(float a, float b) {  // lanes 0-31: a[l]+a[l+32], lanes 32-63: b[l-32]+b[l]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {  // even rows: a folded over row pairs, odd rows: b
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// which accumulator slot (0..11) this lane writes, or -1: lanes 0,8,..,56 hold v[0..7]; lanes 15,31,47,63 the 4 row sums of v[8]
__device__ __forceinline__ int acc_slot(int lane) {
    if ((lane & 7) == 0) {
        const int grp = lane >> 3;  // (row, half): row0 -> v0|v4, row1 -> v2|v6, row2 -> v1|v5, row3 -> v3|v7
        const int row = grp >> 1, half = grp & 1;
        const int base = (row == 0) ? 0 : (row == 1) ? 2 : (row == 2) ? 1 : 3;
        return base + 4 * half;
    }
    if ((lane & 15) == 15) return 8 + (lane >> 4);
    return -1;
}
__device__ __forceinline__ float wave_reduce_scatter9(const float (&v)[9], bool upper8) {
    const float u0 = swap_add32(v[0], v[1]), u1 = swap_add32(v[2], v[3]);
    const float u2 = swap_add32(v[4], v[5]), u3 = swap_add32(v[6], v[7]);
    const float t0 = swap_add16(u0, u1);  // rows: v0, v2, v1, v3
    const float t1 = swap_add16(u2, u3);  // rows: v4, v6, v5, v7
    const float keep = upper8 ? t1 : t0, send = upper8 ? t0 : t1;
    const int sw = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xF, 0xF, true);  // row_ror:8
    float r = keep + __int_as_float(sw);
    r = dpp_add_row<0x141>(r);  // row_half_mirror
    r = dpp_add_row<0x1B>(r);   // quad_perm [3,2,1,0]
    r = dpp_add_row<0xB1>(r);   // quad_perm [1,0,3,2]  -> every lane of an 8-lane group holds its value's wave sum
    float w = v[8];
    w = dpp_add_row<0xB1>(w);
    w = dpp_add_row<0x4E>(w);   // quad_perm [2,3,0,1]
    w = dpp_add_row<0x114>(w);  // row_shr:4
    w = dpp_add_row<0x118>(w);  // row_shr:8 -> lane 15 of each row holds the row sum
    return ((__lane_id() & 15) == 15) ? w : r;
}


}  // namespace
