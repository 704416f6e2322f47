// gsr_composite.hip -- forward and backward alpha compositing for gfx950: ONE wave64 per 8x8-pixel bin.
//
// Semantics: SURVEY.md section 9.2 (front-to-back blend with the power>0 / alpha<1/255 / T<1e-4 rules) and section 9.3 (back-to-
// front gradient recurrence); replaces upstream renderCUDA forward/backward (called through
// /root/reference/gaussian_renderer/__init__.py:54-62 and its autograd backward).
//
// CDNA4 mapping:
//   * lane = pixel of an 8x8 bin; every workgroup is ONE wave, so the dispatcher balances the CUs at wave granularity
//     (4-wave workgroups packed greedily left CUs with 7 busy workgroups next to CUs with 4).  There is no
//     __syncthreads anywhere: a wave that finishes early (all pixels saturated) stops fetching immediately;
//   * per round a wave stages 64 splat records {x,y,A,B | C,op,r,g | b} -- colour included -- with one 48-byte gather
//     per lane into its private LDS slice; the gather for round i+1 is issued BEFORE round i is consumed (registers),
//     so HBM/L2 latency hides under the blend loop; the blend loop reads wave-uniform LDS addresses (broadcast);
//   * lists are exact-extent culled per bin (gsr_common.h), so a wave never iterates a splat that cannot touch it; bins
//     are taken from the work-ordered list in an XCD-aware order (xcd_list_pos) so that neighbouring bins share an L2;
//   * backward: the 9 per-(pixel, splat) gradient terms are summed across the wave with a butterfly reduce-scatter
//     (v_permlane32/16_swap + DPP, 22 instructions instead of 54, no LDS traffic), parked per staged splat in LDS by
//     12 lanes and flushed once per round as one 32-byte record + dL/dopacity + flag per INSTANCE, at the instance's Gaussian-major
//     slot (gsr_common.h).  There is no global atomic in the backward at all (upstream issues 10 per (pixel, splat); float
//     atomics run at only 20-30 Mops/ms on MI355X): k_preprocess_bwd streams each Gaussian's slots in a fixed order, so
//     gradients are also bit-reproducible.
// This file is the VALU-only kernel family (GPSGS_COMPOSITE=valu); the default family takes the exponents from matrix-core tiles
// (gsr_composite_tiles.hip).
#include "gsr_composite_common.h"

namespace {

__global__ __launch_bounds__(64 * WAVES) void k_composite_fwd(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                       GsrBins bins, const uint32_t *__restrict__ wg_order,
                                                       const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                       float *__restrict__ out_color, float *__restrict__ final_T,
                                                       uint32_t *__restrict__ n_contrib, const GsrHeader *__restrict__ hdr, uint8_t *__restrict__ inst_valid) {
    __shared__ float4 sA[WAVES][WAVE];
    __shared__ float4 sB[WAVES][WAVE];
    __shared__ float sC[WAVES][WAVE];
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    const WaveGeom g = wave_geom(W, H, bx, bins, wg_order, list_pos);
    if (hdr->overflow) {  // nothing can be rendered from truncated lists: a deterministic zero image instead of uninitialised memory
        fwd_write_blank(g, W, H, out_color, final_T, n_contrib);
        return;
    }
    clear_record_flags(inst_valid, hdr, (int)threadIdx.x, 64 * WAVES);
    const float pxf = (float)g.px, pyf = (float)g.py;
    float4 *wA = sA[g.wid], *wB = sB[g.wid];
    float *wC = sC[g.wid];

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;      // 1-based list position of the last splat that contributed (n_contrib), up to the previous round
    uint32_t last_rnd = 0;  // ... 1-based slot of the last contributor inside the current round (0: none): an inline constant per select
    // Per-lane predicates live as wave-uniform 64-bit masks in SGPRs and are combined on the scalar unit: one v_cmp per test, never a
    // second compare for the complement; __builtin_amdgcn_inverse_ballot_w64 hands a mask back to v_cndmask for free.
    typedef unsigned long long lanemask_t;
    lanemask_t active = __ballot(g.inside);  // pixels inside the image that are not yet saturated
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r0), r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r1);

    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    if (r0 + g.lane < r1) {  // prefetch round 0
        const float4 *s = reinterpret_cast<const float4 *>(splats + point_list[r0 + g.lane]);
        nA = s[0]; nB = s[1]; nC = s[2].x;
    }
    for (uint32_t base = r0; base < r1; base += WAVE) {
        if (active == 0ull) break;  // every pixel of this bin is saturated (or outside the image)
        wave_sync_lds();            // previous round fully consumed
        wA[g.lane] = make_float4(nA.x, nA.y, -0.5f * GSR_LOG2E * nA.z, -GSR_LOG2E * nA.w);  // conic pre-scaled for gsr_power2
        wB[g.lane] = make_float4(-0.5f * GSR_LOG2E * nB.x, nB.y, nB.z, nB.w);
        wC[g.lane] = nC;
        wave_sync_lds();
        const uint32_t nk = base + WAVE + g.lane;
        nB.y = 0.f;  // a slot without a splat blends nothing (opacity 0 -> alpha 0 < 1/255; stale x, y, conic stay finite)
        if (nk < r1) {  // prefetch the next round while this one is blended
            const float4 *s = reinterpret_cast<const float4 *>(splats + point_list[nk]);
            nA = s[0]; nB = s[1]; nC = s[2].x;
        }
        const int cnt = (int)min((uint32_t)WAVE, r1 - base);
        // Branch-free blend in groups of 8 (the tail group is padded by opacity-0 slots; 4 and 16 measured slower); between groups one
        // scalar test stops the round as soon as all 64 pixels are saturated -- on average half a round (~8 % of a body bin's list) is
        // not walked at all
#pragma unroll
        for (int j0 = 0; j0 < WAVE; j0 += 8) {
            if (j0 < cnt && active != 0ull) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = j0 + u;
                    const float4 a = wA[j];
                    const float4 b = wB[j];
                    const float c2 = wC[j];
                    const float dx = a.x - pxf, dy = a.y - pyf;
                    const float power = gsr_power2(a.z, a.w, b.x, dx, dy);
                    const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
                    const lanemask_t valid = active & ~(__ballot(power > 0.f) | __ballot(alpha < 1.f / 255.f));
                    const float test_T = __builtin_fmaf(-alpha, T, T);  // T (1 - alpha), one rounding
                    const lanemask_t sat = __ballot(test_T < 0.0001f);
                    active &= ~(valid & sat);
                    const bool use = __builtin_amdgcn_inverse_ballot_w64(valid & ~sat);
                    const float w = use ? alpha * T : 0.f;
                    C0 += b.z * w;
                    C1 += b.w * w;
                    C2 += c2 * w;
                    T = use ? test_T : T;
                    last_rnd = use ? (uint32_t)(j + 1) : last_rnd;
                }
            }
        }
        last = last_rnd ? (base - r0) + last_rnd : last;
        last_rnd = 0;
    }
    if (g.inside) {
        const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
        final_T[q] = T;
        n_contrib[q] = last;
        // a view WITHOUT Gaussians is upstream's zero-initialised image, not the background (it skips every kernel when P == 0); with a
        // row range the host does not know the count, so the rule is applied here (wave-uniform scalar load)
        const float bgs = hdr->num_points != 0u ? 1.f : 0.f;
        out_color[q] = C0 + T * (bgs * bg[0]);
        out_color[npix + q] = C1 + T * (bgs * bg[1]);
        out_color[2 * npix + q] = C2 + T * (bgs * bg[2]);
    }
}

__global__ __launch_bounds__(64 * WAVES) void k_composite_bwd(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                       GsrBins bins, const uint32_t *__restrict__ wg_order,
                                                       const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                       const float *__restrict__ dL_dpix, const float *__restrict__ final_T,
                                                       const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ goff,
                                                       const uint32_t *__restrict__ gpart, uint8_t *__restrict__ inst_valid, float *__restrict__ inst_dop,
                                                       GsrGradAcc *__restrict__ inst_grad, const GsrHeader *__restrict__ hdr) {
    __shared__ float4 sA[WAVES][WAVE];
    __shared__ float4 sB[WAVES][WAVE];
    __shared__ float sC[WAVES][WAVE];
    __shared__ float4 sAcc[WAVES][WAVE * 3];  // per staged splat: {dr,dg,db,dmx | dmy,cxx,cxy,cyy | 4 row sums of dop}
    if (hdr->overflow) return;
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    if (list_pos >= hdr->num_busy_wgs) return;  // idle workgroups sit at the end of wg_order
    const WaveGeom g = wave_geom(W, H, bx, bins, wg_order, list_pos);
    if (g.r1 <= g.r0) return;
    const int lane = g.lane;
    const float pxf = (float)g.px, pyf = (float)g.py;
    const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
    float4 *wA = sA[g.wid], *wB = sB[g.wid], *wAcc = sAcc[g.wid];
    float *wC = sC[g.wid], *wAccF = reinterpret_cast<float *>(sAcc[g.wid]);
    const int slot = acc_slot(lane);

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

    float T = T_final, A = 0.f;  // A = (colour accumulated behind the current splat) . dL/dpixel
    const float nTb = -T_final * bg_dot;

    // positions are 0-based from the front of the bin list; walk from max_last-1 down to 0 in rounds of 64
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    uint32_t nRec = 0;  // this lane's staged instance's slot (Gaussian, cell of its bin rect) = index of its gradient record
    const int bin_x = g.bin % bx, bin_y = g.bin / bx;
    auto stage = [&](uint32_t list_pos) {
        const uint32_t id = point_list[list_pos];
        const float4 *s = reinterpret_cast<const float4 *>(splats + id);
        nA = s[0]; nB = s[1];
        const float4 c = s[2];
        nC = c.x;
        const uint32_t lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
        const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff;
        nRec = gpart[id >> GSR_BIN_SHIFT] + goff[id] + (uint32_t)((bin_y - y0) * (x1 - x0) + (bin_x - x0));
    };
    if ((int64_t)lane <= max_last - 1) stage(g.r0 + (uint32_t)(max_last - 1 - lane));
    for (int64_t top = max_last - 1; top >= 0; top -= WAVE) {
        const int cnt = (int)min((int64_t)WAVE, top + 1);
        wave_sync_lds();
        wA[lane] = make_float4(nA.x, nA.y, -0.5f * GSR_LOG2E * nA.z, -GSR_LOG2E * nA.w);  // conic pre-scaled for gsr_power2 (as the forward)
        wB[lane] = make_float4(-0.5f * GSR_LOG2E * nB.x, nB.y, nB.z, nB.w);
        wC[lane] = nC;
        const uint32_t curRec = nRec;
        wave_sync_lds();
        const int64_t ntop = top - WAVE;
        if (ntop - lane >= 0) stage(g.r0 + (uint32_t)(ntop - lane));  // prefetch the next round
        unsigned long long touched = 0ull;  // which staged splats received any gradient (wave-uniform)
        for (int j = 0; j < cnt; j++) {
            const uint32_t pos = (uint32_t)(top - j);
            const float4 a = wA[j];
            const float4 b = wB[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = gsr_power2(a.z, a.w, b.x, dx, dy);
            const float G = __builtin_amdgcn_exp2f(power);
            const float alpha = fminf(0.99f, b.y * G);
            const unsigned long long valid_m = __ballot(pos < last) & ~(__ballot(power > 0.f) | __ballot(alpha < 1.f / 255.f));
            if (valid_m == 0ull) continue;  // wave-uniform
            const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
            touched |= 1ull << j;

            // Per pair only the colour terms and six MOMENTS of s = dL/dG * G are formed: S0 = sum s, Sx = sum s dx, Sy, Sxx,
            // Sxy, Syy.  dL/dmean2D, dL/dconic and dL/dopacity are linear in them and are finished once per (bin, splat)
            // at flush time (11 fewer instructions per pair than forming the nine upstream terms here).
            // Branch-free: a lane this splat does not reach (behind its last contributor, power > 0, alpha < 1/255) runs the
            // same arithmetic with alpha = 0 and G = 0: rcp(1) = 1 leaves T, the recurrence below leaves A (0 * cd + 1 * A), and
            // every sum receives an exact zero.
            // Upstream carries accum_rec (the colour seen behind the splat, 3 channels) with last_alpha / last_color; only its
            // dot product with dL/dpixel is ever used, so the recurrence is carried on that scalar: A <- alpha cd + (1 - alpha) A
            // with cd = colour . dL/dpixel.  Same order of operations back to front (no cancellation), 14 instead of 23
            // instructions, and nothing but T and A is carried from splat to splat.
            const float Ge = valid ? G : 0.f;
            const float ae = valid ? alpha : 0.f;
            const float om = 1.f - ae;
            const float rcp = __builtin_amdgcn_rcpf(om);
            T = T * rcp;
            const float cd = b.z * d0 + b.w * d1 + wC[j] * d2;
            const float w = ae * T;  // dchannel/dcolour
            const float dL_dalpha = (cd - A) * T + nTb * rcp;
            A = ae * cd + om * A;
            const float g_r = w * d0;
            const float g_g = w * d1;
            const float g_b = w * d2;
            const float m_0 = (b.y * dL_dalpha) * Ge;  // s = dL/dG * G, with dL/dG = opacity * dL/dalpha straight through the 0.99 clamp
            const float m_x = m_0 * dx;
            const float m_y = m_0 * dy;
            const float m_xx = m_x * dx;
            const float m_xy = m_x * dy;
            const float m_yy = m_y * dy;
            const float red[9] = {g_r, g_g, g_b, m_x, m_y, m_xx, m_xy, m_yy, m_0};
            const float out = wave_reduce_scatter9(red, (lane & 8) != 0);
            if (slot >= 0) wAccF[12 * j + slot] = out;  // 12 lanes, 12 distinct words of this splat's record
        }
        wave_sync_lds();
        if ((touched >> lane) & 1ull) {  // lane j parks staged splat j's sums as ONE 32-byte instance record + its dL/dopacity (no atomics)
            const float4 v0 = wAcc[3 * lane], v1 = wAcc[3 * lane + 1], rs = wAcc[3 * lane + 2];
            const float4 sa = wA[lane], sb = wB[lane];  // this lane staged splat `lane` itself: conic (sa.z, sa.w, sb.x), opacity sb.y
            const float Sx = v0.w, Sy = v1.x, Sxx = v1.y, Sxy = v1.z, Syy = v1.w;
            const float S0 = (rs.x + rs.y) + (rs.z + rs.w);  // arrives as 4 row sums
            // dG/d(delta) = -G (A dx + B dy), -G (C dy + B dx);  dL/dconic = -0.5 s {dx^2, dx dy, dy^2};  dL/dop = G dL/dalpha = s / op
            // the staged conic is pre-scaled: A = sa.z / (-0.5 log2 e), B = sa.w / (-log2 e), C = sb.x / (-0.5 log2 e)
            const float kA = 2.f / GSR_LOG2E, kB = 1.f / GSR_LOG2E;
            const float g_mx = ddelx_dx * (kA * sa.z * Sx + kB * sa.w * Sy);
            const float g_my = ddely_dy * (kA * sb.x * Sy + kB * sa.w * Sx);
            // the record goes to the instance's SLOT (Gaussian-major; gsr_common.h)
            float4 *dst = reinterpret_cast<float4 *>(inst_grad + curRec);  // one whole 32-byte sector
            dst[0] = make_float4(v0.x, v0.y, v0.z, g_mx);
            dst[1] = make_float4(g_my, -0.5f * Sxx, -0.5f * Sxy, -0.5f * Syy);
            inst_dop[curRec] = S0 * __builtin_amdgcn_rcpf(sb.y);
            inst_valid[curRec] = 1;
        }
    }
}

}  // namespace

void gsr_launch_composite_fwd(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                              const uint32_t *point_list, const float *bg, float *out_color, float *final_T, uint32_t *n_contrib,
                              const GsrHeader *hdr, uint8_t *inst_valid, hipStream_t s) {
    const int wgs = (bx / WAVES) * by;
    if (wgs <= 0) return;
    hipLaunchKernelGGL(k_composite_fwd, dim3(wgs), dim3(64 * WAVES), gsr_debug_lds_pad(), s, W, H, bx, splats, bins, wg_order, point_list, bg, out_color,
                       final_T, n_contrib, hdr, inst_valid);
}

void gsr_launch_composite_bwd(int W, int H, int bx, int by, const GsrSplat *splats, GsrBins bins, const uint32_t *wg_order,
                              const uint32_t *point_list, const float *bg, const float *dL_dpix, const float *final_T,
                              const uint32_t *n_contrib, const uint32_t *goff, const uint32_t *gpart, uint8_t *inst_valid, float *inst_dop,
                              GsrGradAcc *inst_grad, const GsrHeader *hdr, hipStream_t s) {
    const int wgs = (bx / WAVES) * by;
    if (wgs <= 0) return;
    hipLaunchKernelGGL(k_composite_bwd, dim3(wgs), dim3(64 * WAVES), gsr_debug_lds_pad(), s, W, H, bx, splats, bins, wg_order, point_list, bg, dL_dpix,
                       final_T, n_contrib, goff, gpart, inst_valid, inst_dop, inst_grad, hdr);
}
