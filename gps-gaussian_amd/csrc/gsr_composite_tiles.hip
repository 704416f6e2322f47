// gsr_composite_tiles.hip -- forward and backward alpha compositing with the exponents taken from bf16 matrix-core tiles
// (gsr_pow_tiles.h): ONE wave64 per 8x8-pixel bin, lane = pixel, everything else as in gsr_composite.hip.
//
// Semantics: SURVEY.md section 9.2 / 9.3 (upstream renderCUDA forward / backward, reached through
// /root/reference/gaussian_renderer/__init__.py:54-62 and its autograd backward).
//
// Per round of 64 staged splats every lane turns ITS staged splat into 32 bf16 terms (fp64 coefficients about the bin centre, exact
// hi / lo split), four v_permlane32_swap pairs arrange them as MFMA operands, and per half round (32 splats) four
// v_mfma_f32_32x32x16_bf16 produce the exponents of 32 splats x 64 pixels into 32 registers.  The blend loop then starts at
// v_exp_f32: the 8 VALU instructions per (pixel, splat) pair that computed dx, dy and the quadratic form are gone (forward 23 -> 15
// per pair), for ~1.7 instructions per (bin, splat) of operand preparation and 4 cycles per (bin, splat) on the matrix pipe.
// The backward's pair body needs no dx, dy either: its sums over the pixels are formed in a second phase with lane = (splat, pixel row).
#include <atomic>
#include <type_traits>

#include "gsr_pow_tiles.h"
#include "gsr_sort_wave.h"

namespace {

typedef unsigned long long lanemask_t;

// Development aid (gsr_debug_set_wg_trace, tools/wg_trace.py): when set, lane 0 of every compositing workgroup that did work leaves
// {wall clock at start, wall clock at end (100 MHz), shader cycles spent, HW_ID | XCC_ID << 32 | list length << 40} in row blockIdx.x
// (forward) / gridDim.x + blockIdx.x (backward): which SIMD ran which bin when -- the per-SIMD timeline behind the tail / balance numbers
// of DESIGN.md.  NULL in normal operation (one scalar load and one branch per workgroup).
__device__ unsigned long long *g_wg_trace = nullptr;
struct WgTrace {
    unsigned long long *row, w0, c0;
    __device__ __forceinline__ WgTrace(unsigned row_index) {
        unsigned long long *base = g_wg_trace;
        row = base ? base + 4ull * row_index : nullptr;
        w0 = row ? wall_clock64() : 0ull;
        c0 = row ? clock64() : 0ull;
    }
    __device__ __forceinline__ void done(unsigned len) const {
        if (row && threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            row[0] = w0; row[1] = wall_clock64(); row[2] = clock64() - c0;
            row[3] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xffu) << 32) | ((unsigned long long)len << 40);
        }
    }
};

// GSR_FLAG_WAVE_PRIORITY -- hardware wave priority (s_setprio, 0..3) as a scheduling hint; results are unchanged.  A SIMD's arbiter serves
// the highest priority first and the OLDEST wave among equals.  Config 2 has ~5 busy bins (= one-wave workgroups of ~370 list entries)
// per SIMD and a wave on its own is latency-bound (per-workgroup timeline, occupancy sweep, issue probes):
//   forward: all of a SIMD's waves are resident from the start; served oldest first they finish one after the other (40, 43, 48, 53,
//     59 us on one SIMD with five equal lists) and the last one runs alone.  With the priority following the work a wave still has in
//     front of it, relative to the longest list of the view, they finish together and the SIMD stays full to the end;
//   backward: three waves fit per SIMD, the fourth and fifth workgroup start when a slot frees up, next to first-generation waves that
//     are almost done.  Late starters go first: the last three waves of a SIMD then finish together instead of the late starters being
//     what the launch ends on.  (Equalising the first generation as in the forward is worse here: all three slots would free up at once
//     and the second generation would run two-wide.)
__device__ __forceinline__ void prio_by_remaining(uint32_t rem, uint32_t longest) {
    const uint32_t r4 = 4u * rem;  // wave-uniform: scalar compares and branches
    if (r4 >= 3u * longest) __builtin_amdgcn_s_setprio(3);
    else if (r4 >= 2u * longest) __builtin_amdgcn_s_setprio(2);
    else if (r4 >= longest) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void prio_late_starter(uint32_t rem, uint32_t longest) {  // 3 while more than half of the longest list is left, then 2
    if (4u * rem >= 2u * longest) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(2);
}

struct TileFwdState {
    float T, C0, C1, C2;
    uint32_t last_rnd;
    lanemask_t active;
};

// KEEP: the per-pixel state the backward starts from (final T, index of the last contributor) is tracked and written; an inference
// workspace (no backward tail) takes the instantiation without it: one select per pair and 8 bytes per pixel less
template <int S, bool KEEP>
__device__ __forceinline__ void tiles_fwd_half(TileFwdState &st, const f32x16 &d0, const f32x16 &d1, const float4 *__restrict__ wCol, int cnt) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j0 = 32 * S + 8 * q;
        // groups of 8; between groups one scalar test stops the round as soon as all 64 pixels are saturated
        if (j0 < cnt && st.active != 0ull) {
            float p[8];
            pow_group8(d0, d1, q, p);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int j = j0 + e;
                const float4 c = wCol[j];  // {opacity, r, g, b}: wave-uniform address (LDS broadcast)
                // upstream skips power > 0, which its own (PSD) quadratic form never produces.  The exact evaluation here can land one
                // rounding above zero where the true exponent is 0 (splat centre on a pixel centre): such a pair is blended (exp2 of
                // +1e-7 is 1 to fp32 precision), never skipped
                const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(p[e]));  // the tile holds log2(opacity G)
                const lanemask_t valid = st.active & ~__ballot(alpha < 1.f / 255.f);
                const float test_T = __builtin_fmaf(-alpha, st.T, st.T);  // T (1 - alpha), one rounding
                const lanemask_t sat = __ballot(test_T < 0.0001f);
                st.active &= ~(valid & sat);
                const bool use = __builtin_amdgcn_inverse_ballot_w64(valid & ~sat);
                // the blend weight alpha T as the transmittance the splat takes away, T - T (1 - alpha): one subtraction after the
                // select (exactly 0 for a pair that is not blended) instead of a product and a second select.  It differs from the
                // rounded product by at most 1 ulp of T (6e-8 T absolute) -- the size of the accumulation's own rounding
                const float T_next = use ? test_T : st.T;
                const float w = st.T - T_next;
                st.C0 += c.y * w;
                st.C1 += c.z * w;
                st.C2 += c.w * w;
                st.T = T_next;
                if (KEEP) st.last_rnd = use ? (uint32_t)(j + 1) : st.last_rnd;
            }
        }
    }
}

#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 5  // waves per SIMD asked of the compiler (92 VGPRs fit 5)
#endif
// SORT (direct lists): the wave first sorts its own bin's list -- keys[r0, r1) -> point_list[r0, r1), one wave per list of <= 1,024 keys exactly as
// k_sort_wave does it (gsr_sort_wave.h) -- and then walks it: no sort launch in front of the compositing.  The sorted ids go to global memory for the
// backward and stay in LDS for this walk (first form: read back from global memory behind a wait for the wave's own stores).
template <bool KEEP, bool SORT>
__global__ __launch_bounds__(64, GSR_FWD_WAVES) void k_composite_fwd_tiles(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                            GsrBins bins, const uint32_t *__restrict__ wg_order,
                                                            uint32_t *point_list, const float *__restrict__ bg,
                                                            float *__restrict__ out_color, float *__restrict__ final_T,
                                                            uint32_t *__restrict__ n_contrib, const GsrHeader *__restrict__ hdr, uint8_t *__restrict__ inst_valid,
                                                            int wave_prio, const uint64_t *__restrict__ keys) {
    __shared__ float4 sCol[WAVE];  // {opacity, r, g, b} of the 64 staged splats
    const WgTrace trace(blockIdx.x);
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    const WaveGeom g = wave_geom(W, H, bx, bins, wg_order, list_pos);
    if (hdr->overflow) {  // nothing can be rendered from truncated lists: a deterministic zero image instead of uninitialised memory
        fwd_write_blank(g, W, H, out_color, final_T, n_contrib);
        return;
    }
    if (KEEP) clear_record_flags(inst_valid, hdr, g.lane, WAVE);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r0), r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r1);
    const int lane = g.lane;
    __shared__ uint32_t sIds[SORT ? 1024 : 1];
    if (SORT && r1 > r0) {  // (wave-uniform)
        if (wave_prio) prio_by_remaining(r1 - r0, hdr->max_tile_count);  // the longest lists sort first and start their walks while the short ones still sort
        sort_wave_list(keys + r0, r1 - r0, point_list + r0, sIds, lane);  // the sorted ids: to global memory for the backward, and in sIds for this walk
        sort_sync(true);
    }
    const float cx = (float)(g.px - (lane & 7)) + 3.5f, cy = (float)(g.py - (lane >> 3)) + 3.5f;  // bin centre
    const PowOperandsB opB = pow_operands_b(lane);

    TileFwdState st;
    st.T = 1.f; st.C0 = 0.f; st.C1 = 0.f; st.C2 = 0.f;
    st.last_rnd = 0;
    st.active = __ballot(g.inside);
    uint32_t last = 0;  // 1-based list position of the last splat that contributed (n_contrib)

    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    if (r0 + lane < r1) {  // prefetch round 0
        const float4 *s = reinterpret_cast<const float4 *>(splats + (SORT ? sIds[lane] : point_list[r0 + lane]));
        nA = s[0]; nB = s[1]; nC = s[2].x;
    }
    for (uint32_t base = r0; base < r1; base += WAVE) {
        if (st.active == 0ull) break;  // every pixel of this bin is saturated (or outside the image)
        if (wave_prio) prio_by_remaining(r1 - base, hdr->max_tile_count);
        const PowOperandsA opA = pow_operands_a(pow_terms(nA.x, nA.y, nA.z, nA.w, nB.x, cx, cy, gsr_log2_opacity(nB.y)));
        wave_sync_lds();  // previous round fully consumed
        sCol[lane] = make_float4(nB.y, nB.z, nB.w, nC);
        wave_sync_lds();
        const uint32_t nk = base + WAVE + lane;
        nB.y = 0.f;  // a slot without a splat blends nothing (opacity 0 -> alpha 0 < 1/255; stale x, y, conic stay finite)
        if (nk < r1) {  // prefetch the next round while this one is blended
            const float4 *s = reinterpret_cast<const float4 *>(splats + (SORT ? sIds[nk - r0] : point_list[nk]));
            nA = s[0]; nB = s[1]; nC = s[2].x;
        }
        const int cnt = (int)min((uint32_t)WAVE, r1 - base);
        {
            const f32x16 d0 = pow_tile_bf16(opA.a[0], opB.b[0]), d1 = pow_tile_bf16(opA.a[0], opB.b[1]);
            tiles_fwd_half<0, KEEP>(st, d0, d1, sCol, cnt);
        }
        if (cnt > 32 && st.active != 0ull) {
            const f32x16 d0 = pow_tile_bf16(opA.a[1], opB.b[0]), d1 = pow_tile_bf16(opA.a[1], opB.b[1]);
            tiles_fwd_half<1, KEEP>(st, d0, d1, sCol, cnt);
        }
        if (KEEP) {
            last = st.last_rnd ? (base - r0) + st.last_rnd : last;
            st.last_rnd = 0;
        }
    }
    if (g.inside) {
        const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
        if (KEEP) {
            final_T[q] = st.T;
            n_contrib[q] = last;
        }
        // a view WITHOUT Gaussians is upstream's zero-initialised image, not the background (it skips every kernel when P == 0); with a
        // row range the host does not know the count, so the rule is applied here (wave-uniform scalar load)
        const float bgs = hdr->num_points != 0u ? 1.f : 0.f;
        out_color[q] = st.C0 + st.T * (bgs * bg[0]);
        out_color[npix + q] = st.C1 + st.T * (bgs * bg[1]);
        out_color[2 * npix + q] = st.C2 + st.T * (bgs * bg[2]);
    }
    if (r1 > r0) trace.done(r1 - r0);
}

// ---- backward: scalar colour-behind recurrence and atomic-free per-instance records as in gsr_composite.hip, exponents from the tiles,
// and the nine per-(bin, splat) sums over the 64 pixels taken in TWO PHASES instead of a 64-lane butterfly per splat:
//   phase 1 (lane = pixel, 8 staged splats): the per-pixel recurrences; each (pixel, splat) pair leaves just two numbers in LDS:
//       s = dL/dG * G   and   w = alpha * T (dchannel/dcolour);
//   phase 2 (lane = (splat j = lane & 7, pixel row = lane >> 3)): the lane reads the 8 pixels of its row for its splat (two
//       ds_read_b128 per array, XOR-swizzled: conflict-free on both sides), forms the colour sums and the moments of s about the
//       splat centre with every lane busy, and a 3-level reduce-scatter over the 8 rows (permlane32 / permlane16 swaps, one DPP
//       rotation) finishes 8 splats at once: ~11 VALU instructions per (bin, splat) instead of 9 products + 22 of butterfly,
//       and the pair body no longer needs dx, dy at all.
struct TileBwdState {
    float T, A;
    lanemask_t touched;
};
struct TileBwdConst {
    float *wr[4];        // phase 1: this lane's word of a splat's 64-pixel row, for the four swizzles f(e) = 4 (e & 1) + 32 ((e >> 1) & 1)
    const float4 *rdA;   // phase 2: pixels u = 0..3 of (splat j, row); rdB: u = 4..7
    const float4 *rdB;
    float *acc;          // phase 2: &wAccF[12 * j + slot(row group)]
    float *acc0;         // phase 2: &wAccF[12 * j + 8 + (lane >> 4)]  (the four 16-lane partial sums of S0)
    float pyrow;         // pixel-centre y of this lane's phase-2 row
    float dr[8], dg[8], db[8];  // dL/dpixel of the 8 pixels of that row
};
constexpr int TILE_SW_WORDS = 8 * WAVE;  // one array (s or w): 8 splats x 64 pixels

// CG = false: nobody asked for dL/dcolour (GSR_FLAG_NO_COLOR_GRAD -- stage 2: the colours are input pixels, train_stage2.py never
// differentiates them): the w array is neither written nor read and the three colour sums per (pixel, splat) are left out; their slots of the
// reduce-scatter carry zeros, so the record layout and every other gradient are bit-identical to the full form
template <int GQ, bool CG>
__device__ __forceinline__ void tiles_bwd_phase2(const TileBwdConst &k, const float4 *__restrict__ wXY, const float (&pxu)[8], int lane) {
    const float4 s0 = k.rdA[0], s1 = k.rdB[0];
    const float2 xy = *reinterpret_cast<const float2 *>(&wXY[8 * GQ + (lane & 7)]);
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    // the sums start from the u = 0 terms (0 + x and fma(a, b, 0) are not folded by the compiler: -0 semantics; same bits)
    const float dx0 = xy.x - pxu[0], mx0 = sv[0] * dx0;
    float S0 = sv[0], Sx = mx0, Sxx = mx0 * dx0, cr = 0.f, cg = 0.f, cb = 0.f;
    if (CG) {
        const float4 w0 = k.rdA[TILE_SW_WORDS / 4], w1 = k.rdB[TILE_SW_WORDS / 4];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        cr = wv[0] * k.dr[0]; cg = wv[0] * k.dg[0]; cb = wv[0] * k.db[0];
#pragma unroll
        for (int u = 1; u < 8; u++) {
            cr = __builtin_fmaf(wv[u], k.dr[u], cr);
            cg = __builtin_fmaf(wv[u], k.dg[u], cg);
            cb = __builtin_fmaf(wv[u], k.db[u], cb);
        }
    }
#pragma unroll
    for (int u = 1; u < 8; u++) {
        const float dx = xy.x - pxu[u];  // the same single rounding as x - (float)px
        const float mx = sv[u] * dx;
        S0 += sv[u];
        Sx += mx;
        Sxx = __builtin_fmaf(mx, dx, Sxx);
    }
    const float dy = xy.y - k.pyrow;  // constant along the row
    const float Sy = dy * S0, Sxy = dy * Sx;
    const float Syy = dy * Sy;
    // reduce-scatter over the 8 rows (lane bits 5, 4, 3); lane bits 0..2 (the splat) are preserved by every step
    const float u0 = swap_add32(cr, cg), u1 = swap_add32(cb, Sx), u2 = swap_add32(Sy, Sxx), u3 = swap_add32(Sxy, Syy);
    const float t0 = swap_add16(u0, u1), t1 = swap_add16(u2, u3);
    const bool upper8 = (lane & 8) != 0;
    const float keep = upper8 ? t1 : t0, send = upper8 ? t0 : t1;
    const float r = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xF, 0xF, true));  // row_ror:8
    const float z = dpp_add_row<0x128>(S0);
    k.acc[12 * 8 * GQ] = r;
    if (!upper8) k.acc0[12 * 8 * GQ] = z;
}

// one staged splat of phase 1; inr = lanes whose pixel has this list position among its contributors.  BG = false: the background is
// black (the reference's stage 2, config/stereo_human_config.py:24), its term of dL/dalpha is an exact zero and is left out
template <bool BG, bool CG>
__device__ __forceinline__ void tiles_bwd_pair(TileBwdState &st, const TileBwdConst &k, const float4 c, float pe, lanemask_t inr, int e, int j,
                                               float dr, float dg, float db, float nTb) {
    const float aG = __builtin_amdgcn_exp2f(pe);  // opacity * G (the tile holds its log2); as the forward: no power > 0 skip
    // the forward's test alpha = min(0.99, aG) < 1/255 is aG < 1/255: the clamp is applied after the select (one select instead of two)
    const lanemask_t valid_m = inr & ~__ballot(aG < 1.f / 255.f);
    // (Tried: no wave-uniform skip -- the entry's numbers are exact zeros either way, and the branch sits on a VALU -> SALU round trip:
    //  111.4 vs 111.4 us with the chip to itself, -1.5 % with six views in flight: kept.)
    if (valid_m != 0ull) {  // wave-uniform
        st.touched |= 1ull << j;
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
        const float cd = c.x * dr + c.y * dg + c.z * db;
        const float aGe = valid ? aG : 0.f;  // dalpha/dG * G = opacity * G, straight through the 0.99 clamp
        float ae;  // min(0.99, aGe) as ONE v_min_f32 (fminf() first canonicalises the select's result: a wasted v_max)
        __asm__("v_min_f32_e32 %0, 0x3f7d70a4, %1" : "=v"(ae) : "v"(aGe));
        const float om = 1.f - ae;
        const float rcp = __builtin_amdgcn_rcpf(om);
        st.T = st.T * rcp;
        const float cA = cd - st.A;
        const float dL_dalpha = BG ? cA * st.T + nTb * rcp : cA * st.T;
        st.A = __builtin_fmaf(ae, cA, st.A);  // = ae cd + (1 - ae) A
        k.wr[e & 3][64 * e] = dL_dalpha * aGe;             // s = dL/dG * G
        if (CG) k.wr[e & 3][TILE_SW_WORDS + 64 * e] = ae * st.T;   // w = dchannel/dcolour
    }
}

template <int GQ, bool BG, bool CG>
__device__ __forceinline__ void tiles_bwd_group(TileBwdState &st, const f32x16 &d0, const f32x16 &d1, const float4 *__restrict__ wXY,
                                                const float4 *__restrict__ wCol, const TileBwdConst &k, const float (&pxu)[8], int lane,
                                                uint32_t topu, uint32_t last, float dr, float dg, float db, float nTb) {
    float p[8];
    pow_group8(d0, d1, GQ & 3, p);
    // {r, g, b} of staged splat j + 1 is fetched from LDS while splat j is processed (the load sits ahead of the wave-uniform
    // skip branch and is consumed after it).  Staged slot j sits at list position top - j (slots beyond the front of the list: topu - j
    // wraps to a huge position, and they carry opacity 0).
    // (Tried: one in-range mask per group of 8 when its first and last position select the same lanes -- saves a compare per pair but
    //  needs a second copy of the group body: 128 -> 159 us, instruction-cache bound.)
    float4 cn = wCol[8 * GQ];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float4 c = cn;
        if (e < 7) cn = wCol[8 * GQ + e + 1];
        tiles_bwd_pair<BG, CG>(st, k, c, p[e], __ballot(last > topu - (uint32_t)(8 * GQ + e)), e, 8 * GQ + e, dr, dg, db, nTb);
    }
    // splats of the group that no pixel touched leave stale numbers in their rows: their sums are never flushed (touched bit clear)
    if ((st.touched >> (8 * GQ)) & 0xffull) tiles_bwd_phase2<GQ, CG>(k, wXY, pxu, lane);
}

// ~170 VGPRs: 2 waves per SIMD (2 and 3 measured identical: the kernel is VALU-issue bound; forcing 4 spills: 172 us)
// The instantiation WITHOUT dL/dcolour (GSR_FLAG_NO_COLOR_GRAD: the gradient set stage 2 differentiates) needs 139 VGPRs on its own = 3 waves per SIMD; asked for 4
// the compiler fits it into 128 with 3 spilled dwords.  Where the kernel is THROUGHPUT-bound -- ~19 waves per SIMD, 95 % SIMD utilisation: config 2 rendered at 2048^2
// (what stage 2 renders), config 5, the large-splat regime (profiles/r04_regime_wg_timeline.md) -- the fourth wave pays: 183.7 -> 165.4 / 587.4 -> 554.9 / 669.0 -> 627.0 us;
// at config 2 (five work items per SIMD, latency-bound) nothing changes (106.7 -> 107.1).  Same instructions, same results.
#ifndef GSR_BWD_NOCOLOR_WAVES
#define GSR_BWD_NOCOLOR_WAVES 4
#endif
template <bool CG>
__global__ __launch_bounds__(64, CG ? 2 : GSR_BWD_NOCOLOR_WAVES) void k_composite_bwd_tiles(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                            GsrBins bins, const uint32_t *__restrict__ wg_order,
                                                            const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                            const float *__restrict__ dL_dpix, const float *__restrict__ final_T,
                                                            const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ goff,
                                                            const uint32_t *__restrict__ gpart, uint8_t *__restrict__ inst_valid, float *__restrict__ inst_dop,
                                                            GsrGradAcc *__restrict__ inst_grad,
                                                            const GsrHeader *__restrict__ hdr, uint32_t prio_from_wg) {
    __shared__ float4 sXY[WAVE];       // {x, y, A, B} of the staged splats (x, y for the moments; A, B for the flush)
    __shared__ float4 sCol[WAVE];      // {r, g, b, -}: colour first, so that one aligned ds_read_b96 (immediate offset) fetches it
    __shared__ float4 sAcc[WAVE * 3];  // per staged splat: {dr,dg,db,Sx | Sy,Sxx,Sxy,Syy | 4 partial sums of S0}
    __shared__ float4 sSW[2 * TILE_SW_WORDS / 4];  // phase 1 -> phase 2: s[8 splats][64 pixels], w[8][64], rows XOR-swizzled
    if (hdr->overflow) return;
    const WgTrace trace(gridDim.x + blockIdx.x);
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    if (list_pos >= hdr->num_busy_wgs) return;  // idle workgroups sit at the end of wg_order
    const WaveGeom g = wave_geom(W, H, bx, bins, wg_order, list_pos);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r0), r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r1);
    if (r1 <= r0) return;
    const int lane = g.lane;
    const float pxf = (float)g.px, pyf = (float)g.py;
    const float cx = pxf - (float)(lane & 7) + 3.5f, cy = pyf - (float)(lane >> 3) + 3.5f;  // bin centre
    const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
    float *wAccF = reinterpret_cast<float *>(sAcc);

    const float T_final = g.inside ? final_T[q] : 0.f;
    const uint32_t last = g.inside ? n_contrib[q] : 0u;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (g.inside) {
        d0 = dL_dpix[q];
        d1 = dL_dpix[npix + q];
        d2 = dL_dpix[2 * npix + q];
    }
    const float bg_dot = bg[0] * d0 + bg[1] * d1 + bg[2] * d2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // deepest contributor over the bin: nothing behind it receives gradient
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    const int64_t max_last = (int64_t)__builtin_amdgcn_readfirstlane((int)m);
    if (max_last == 0) return;
    const PowOperandsB opB = pow_operands_b(lane);

    // phase-2 constants of this lane = (splat j = lane & 7 of a group of 8, pixel row = lane >> 3)
    TileBwdConst k;
    float pxu[8];  // pixel-centre x of the bin's 8 columns: wave-uniform (scalar registers)
    {
        float *sw = reinterpret_cast<float *>(sSW);
        const int j = lane & 7, row = lane >> 3;
#pragma unroll
        for (int f = 0; f < 4; f++) k.wr[f] = sw + (lane ^ (4 * (f & 1) + 32 * (f >> 1)));
        const int fj = 4 * (j & 1) + 32 * ((j >> 1) & 1);
        k.rdA = reinterpret_cast<const float4 *>(sw + 64 * j + ((8 * row) ^ fj));
        k.rdB = reinterpret_cast<const float4 *>(sw + 64 * j + ((8 * row + 4) ^ fj));
        const int grp = lane >> 3, r4 = grp >> 1;  // reduce-scatter placement as acc_slot(): row0 -> v0|v4, row1 -> v2|v6, row2 -> v1|v5, row3 -> v3|v7
        k.acc = wAccF + 12 * j + ((r4 == 0) ? 0 : (r4 == 1) ? 2 : (r4 == 2) ? 1 : 3) + 4 * (grp & 1);
        k.acc0 = wAccF + 12 * j + 8 + (lane >> 4);
        k.pyrow = pyf - (float)(lane >> 3) + (float)row;  // = this lane's own pixel row
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int src = (lane & 56) + u;  // the lane whose pixel is (row, u)
            k.dr[u] = CG ? __shfl(d0, src, 64) : 0.f;
            k.dg[u] = CG ? __shfl(d1, src, 64) : 0.f;
            k.db[u] = CG ? __shfl(d2, src, 64) : 0.f;
            pxu[u] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pxf - (float)(lane & 7) + (float)u)));
        }
    }

    TileBwdState st;
    st.T = T_final; st.A = 0.f; st.touched = 0ull;
    const float nTb = -T_final * bg_dot;

    // positions are 0-based from the front of the bin list; walk from max_last-1 down to 0 in rounds of 64
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    uint32_t nRec = 0;  // this lane's staged instance's slot (Gaussian, cell of its bin rect) = index of its gradient record
    const int bin_x = g.bin % bx, bin_y = g.bin / bx;
    auto stage = [&](int64_t pos) {
        nB.y = 0.f;  // a slot without a splat blends nothing (opacity 0)
        if (pos >= 0) {
            const uint32_t id = point_list[r0 + (uint32_t)pos];
            const float4 *s = reinterpret_cast<const float4 *>(splats + id);
            nA = s[0]; nB = s[1];
            const float4 c = s[2];
            nC = c.x;
            const uint32_t lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
            const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff;
            nRec = gpart[id >> GSR_BIN_SHIFT] + goff[id] + (uint32_t)((bin_y - y0) * (x1 - x0) + (bin_x - x0));
        }
    };
    stage(max_last - 1 - lane);
    // the walk exists twice (generic lambda, instantiated per background class); one launch only ever runs one of the two
    auto walk = [&](auto bg_tag) {
    constexpr bool BG = decltype(bg_tag)::value;
    for (int64_t top = max_last - 1; top >= 0; top -= WAVE) {
        const int cnt = (int)min((int64_t)WAVE, top + 1);
        const uint32_t topu = (uint32_t)top;
        // GSR_FLAG_WAVE_PRIORITY: workgroups behind the first resident generation (prio_from_wg = what the chip holds at once; 0 = off)
        if (prio_from_wg && blockIdx.x >= prio_from_wg) prio_late_starter(topu + 1u, hdr->max_tile_count);
        const PowOperandsA opA = pow_operands_a(pow_terms(nA.x, nA.y, nA.z, nA.w, nB.x, cx, cy, gsr_log2_opacity(nB.y)));
        wave_sync_lds();  // previous round fully consumed
        sXY[lane] = nA;
        sCol[lane] = make_float4(nB.z, nB.w, nC, 0.f);
        const float sC_ = nB.x, sop = nB.y;  // this lane's own staged splat, for the flush
        const uint32_t curRec = nRec;
        wave_sync_lds();
        stage(top - WAVE - lane);  // prefetch the next round
        st.touched = 0ull;
        {
            const f32x16 t0 = pow_tile_bf16(opA.a[0], opB.b[0]), t1 = pow_tile_bf16(opA.a[0], opB.b[1]);
            if (0 < cnt) tiles_bwd_group<0, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (8 < cnt) tiles_bwd_group<1, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (16 < cnt) tiles_bwd_group<2, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (24 < cnt) tiles_bwd_group<3, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
        }
        if (32 < cnt) {
            const f32x16 t0 = pow_tile_bf16(opA.a[1], opB.b[0]), t1 = pow_tile_bf16(opA.a[1], opB.b[1]);
            tiles_bwd_group<4, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (40 < cnt) tiles_bwd_group<5, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (48 < cnt) tiles_bwd_group<6, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
            if (56 < cnt) tiles_bwd_group<7, BG, CG>(st, t0, t1, sXY, sCol, k, pxu, lane, topu, last, d0, d1, d2, nTb);
        }
        wave_sync_lds();
        if ((st.touched >> lane) & 1ull) {  // lane j parks staged splat j's sums as ONE 32-byte instance record + its dL/dopacity (no atomics)
            const float4 v0 = sAcc[3 * lane], v1 = sAcc[3 * lane + 1], rs = sAcc[3 * lane + 2];
            const float4 sa = sXY[lane];  // this lane staged splat `lane` itself: conic A = sa.z, B = sa.w, C = sC_
            const float Sx = v0.w, Sy = v1.x, Sxx = v1.y, Sxy = v1.z, Syy = v1.w;
            const float S0 = (rs.x + rs.y) + (rs.z + rs.w);  // arrives as 4 row sums
            // dG/d(delta) = -G (A dx + B dy), -G (C dy + B dx);  dL/dconic = -0.5 s {dx^2, dx dy, dy^2};  dL/dop = G dL/dalpha = s / op
            const float g_mx = ddelx_dx * (-sa.z * Sx - sa.w * Sy);
            const float g_my = ddely_dy * (-sC_ * Sy - sa.w * Sx);
            // the record goes to the instance's SLOT (Gaussian-major: scattered stores here, a streaming read in k_preprocess_bwd;
            // scattered reads are what costs on this memory system, gsr_common.h)
            float4 *dst = reinterpret_cast<float4 *>(inst_grad + curRec);  // one whole 32-byte sector
            const float dop = S0 * __builtin_amdgcn_rcpf(sop);
            if (CG) {
                dst[0] = make_float4(v0.x, v0.y, v0.z, g_mx);
                inst_dop[curRec] = dop;
            } else {
                // without the colour sums three floats of the record are dead: dL/dopacity rides in the first of them -- the record is then ONE
                // sector instead of a sector + a 4-byte store into inst_dop that costs a sector of its own (round 5 counters, untrained-heads
                // regime: 98 bytes written per 37-byte record), and k_preprocess_bwd gathers one line per record instead of two
                dst[0] = make_float4(dop, 0.f, 0.f, g_mx);
            }
            dst[1] = make_float4(g_my, -0.5f * Sxx, -0.5f * Sxy, -0.5f * Syy);
            inst_valid[curRec] = 1;
        }
    }
    };
    const bool black = __builtin_amdgcn_readfirstlane((int)(bg[0] == 0.f && bg[1] == 0.f && bg[2] == 0.f)) != 0;
    if (black) walk(std::false_type{});
    else walk(std::true_type{});
    trace.done((unsigned)max_last);
}

// ---- device self-test of the tile plumbing (gsr_selftest): the SAME device functions the kernels use, on pseudo-random splats
// around a bin, against the quadratic form evaluated per lane in fp64.  out[0] = max |tile - fp64| / (1 + |value|) over 64 splats x 64
// pixels, out[1] = the same restricted to pairs with exponent > -12 natural units (the ones that can pass the alpha test),
// out[2] = 1 if v_permlane32_swap behaves as documented, out[3] = largest |c0| met (log2 units).
__global__ __launch_bounds__(64) void k_selftest_tiles(float *__restrict__ out) {
    __shared__ float sP[WAVE][WAVE + 1];  // [splat][pixel] exponents
    __shared__ float sRec[WAVE][5];
    const int lane = threadIdx.x;
    auto rnd = [](uint32_t a) {
        a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
        return (float)(a & 0xffffff) * (1.f / 16777216.f);
    };
    uint32_t lo, hi;
    gsr_swap32((uint32_t)lane, (uint32_t)(lane + 100), lo, hi);
    const bool swap_ok = (lane < 32 ? (lo == (uint32_t)lane && hi == (uint32_t)(lane + 32)) : (lo == (uint32_t)(lane + 100 - 32) && hi == (uint32_t)(lane + 100)));
    const float cx = 515.5f, cy = 259.5f;
    // centres up to ~15 px from the bin centre, conic eigenvalues up to 1/0.3 (the dilation bound), arbitrary orientation
    const float x = cx + (rnd(lane * 7 + 1) - 0.5f) * 30.f, y = cy + (rnd(lane * 7 + 2) - 0.5f) * 30.f;
    const float a = 0.02f + 3.3f * rnd(lane * 7 + 3), c = 0.02f + 3.3f * rnd(lane * 7 + 4), b = (rnd(lane * 7 + 5) - 0.5f) * 1.9f * sqrtf(a * c);
    sRec[lane][0] = x; sRec[lane][1] = y; sRec[lane][2] = a; sRec[lane][3] = b; sRec[lane][4] = c;
    const PowOperandsA opA = pow_operands_a(pow_terms(x, y, a, b, c, cx, cy, 0.f));
    const PowOperandsB opB = pow_operands_b(lane);
#pragma unroll
    for (int S = 0; S < 2; S++) {
        const f32x16 d0 = pow_tile_bf16(opA.a[S], opB.b[0]), d1 = pow_tile_bf16(opA.a[S], opB.b[1]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float p[8];
            pow_group8(d0, d1, q, p);
#pragma unroll
            for (int e = 0; e < 8; e++) sP[32 * S + 8 * q + e][lane] = p[e];
        }
    }
    __syncthreads();
    float e0 = 0.f, e1 = 0.f, c0max = 0.f;
    const double px = (double)cx - 3.5 + (lane & 7), py = (double)cy - 3.5 + (lane >> 3);
    for (int j = 0; j < WAVE; j++) {
        const double dx = (double)sRec[j][0] - px, dy = (double)sRec[j][1] - py;
        const double nat = -0.5 * ((double)sRec[j][2] * dx * dx + (double)sRec[j][4] * dy * dy) - (double)sRec[j][3] * dx * dy;
        const double ref = nat * GSR_LOG2E_D;
        const float err = (float)(fabs((double)sP[j][lane] - ref) / (1.0 + fabs(ref)));
        e0 = fmaxf(e0, err);
        if (nat > -12.0) e1 = fmaxf(e1, err);
        const double X = (double)sRec[j][0] - cx, Y = (double)sRec[j][1] - cy;
        c0max = fmaxf(c0max, (float)fabs(0.5 * GSR_LOG2E_D * ((double)sRec[j][2] * X * X + 2.0 * sRec[j][3] * X * Y + (double)sRec[j][4] * Y * Y)));
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        e0 = fmaxf(e0, __shfl_xor(e0, dd, 64)); e1 = fmaxf(e1, __shfl_xor(e1, dd, 64)); c0max = fmaxf(c0max, __shfl_xor(c0max, dd, 64));
    }
    const unsigned long long okm = __ballot(swap_ok);
    if (lane == 0) { out[0] = e0; out[1] = e1; out[2] = (okm == ~0ull) ? 1.f : 0.f; out[3] = c0max; }
}

}  // namespace

void gsr_launch_composite_fwd_tiles(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                                    uint32_t *point_list, const float *bg, float *out_color, float *final_T, uint32_t *n_contrib,
                                    const GsrHeader *hdr, uint8_t *inst_valid, bool keep_state, bool wave_prio, const uint64_t *unsorted_keys, hipStream_t s) {
    const int wgs = bx * by;
    if (wgs <= 0) return;
#define GSR_FWD_TILES_LAUNCH(KEEP_, SORT_)                                                                                                                       \
    hipLaunchKernelGGL((k_composite_fwd_tiles<KEEP_, SORT_>), dim3(wgs), dim3(64), gsr_debug_lds_pad(), s, W, H, bx, splats, bins, wg_order, point_list, bg, out_color, \
                       final_T, n_contrib, hdr, KEEP_ ? inst_valid : nullptr, wave_prio ? 1 : 0, unsorted_keys)
    if (keep_state) { if (unsorted_keys) GSR_FWD_TILES_LAUNCH(true, true); else GSR_FWD_TILES_LAUNCH(true, false); }
    else { if (unsorted_keys) GSR_FWD_TILES_LAUNCH(false, true); else GSR_FWD_TILES_LAUNCH(false, false); }
#undef GSR_FWD_TILES_LAUNCH
}

void gsr_launch_composite_bwd_tiles(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                                    const uint32_t *point_list, const float *bg, const float *dL_dpix, const float *final_T,
                                    const uint32_t *n_contrib, const uint32_t *goff, const uint32_t *gpart, uint8_t *inst_valid, float *inst_dop,
                                    GsrGradAcc *inst_grad, const GsrHeader *hdr, bool color_grad, bool wave_prio, hipStream_t s) {
    const int wgs = bx * by;
    if (wgs <= 0) return;
    // workgroups the chip holds at once (the first generation): occupancy x compute units, asked once per (device, kernel instantiation) and kept in
    // relaxed atomics (a process may drive several GPUs from several host threads; a lost race only asks the runtime twice).  The LDS pad is a
    // process-wide development knob (gsr_debug_lds_pad), so it is not part of the key.
    static std::atomic<uint32_t> resident[64][2];
    uint32_t res = 0u;
    if (wave_prio) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
            res = 0xffffffffu;
        } else if ((res = resident[dev][color_grad ? 1 : 0].load(std::memory_order_relaxed)) == 0u) {
            const void *fn = color_grad ? reinterpret_cast<const void *>(k_composite_bwd_tiles<true>) : reinterpret_cast<const void *>(k_composite_bwd_tiles<false>);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, gsr_debug_lds_pad()) == hipSuccess &&
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
                res = (uint32_t)per_cu * (uint32_t)cus;
            else
                res = 0xffffffffu;  // unknown: no workgroup counts as a late starter
            resident[dev][color_grad ? 1 : 0].store(res, std::memory_order_relaxed);
        }
    }
    const uint32_t prio_from_wg = wave_prio ? res : 0u;
    if (color_grad)
        hipLaunchKernelGGL(k_composite_bwd_tiles<true>, dim3(wgs), dim3(64), gsr_debug_lds_pad(), s, W, H, bx, splats, bins, wg_order, point_list, bg,
                           dL_dpix, final_T, n_contrib, goff, gpart, inst_valid, inst_dop, inst_grad, hdr, prio_from_wg);
    else
        hipLaunchKernelGGL(k_composite_bwd_tiles<false>, dim3(wgs), dim3(64), gsr_debug_lds_pad(), s, W, H, bx, splats, bins, wg_order, point_list, bg,
                           dL_dpix, final_T, n_contrib, goff, gpart, inst_valid, inst_dop, inst_grad, hdr, prio_from_wg);
}

int gsr_set_wg_trace(unsigned long long *rows_device) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &rows_device, sizeof(rows_device)) == hipSuccess ? 0 : -1;
}

void gsr_launch_selftest(float *out, hipStream_t s) { hipLaunchKernelGGL(k_selftest_tiles, dim3(1), dim3(64), 0, s, out); }
