// gsr_composite_mfma.hip -- forward and backward alpha compositing for gfx950 with the matrix cores doing the two parts of the
// per-(pixel, splat) work that are (bi)linear: ONE wave64 per 8x8-pixel bin, lane = pixel (as gsr_composite.hip).
//
// Semantics: SURVEY.md section 9.2 / 9.3 (upstream renderCUDA forward / backward, reached through
// /root/reference/gaussian_renderer/__init__.py:54-62 and its autograd backward).
//
// Why matrix cores in a rasteriser.  Both compositing kernels are bound by VALU ISSUE (one wave64 instruction holds a SIMD for
// 4 cycles), not by HBM, and the MFMA pipe of every SIMD sits idle next to them.  Two pieces of the
// per-pair arithmetic are matrix products and move there:
//
//   (1) the exponent.  power(pixel, splat) = -1/2 d^T Conic d is a quadratic polynomial in the pixel coordinates (u, v) measured
//       from the bin centre:  power = c0 + cu u + cv v + cuu u^2 + cuv u v + cvv v^2  with six per-splat coefficients -- i.e.
//       POWER[splat, pixel] = COEF[splat, 6] x MONO[6, pixel].  Evaluated naively in fp32 that expansion cancels (|c0| can be
//       100 where the value is -5), so every coefficient is formed in fp64 once per (bin, splat) and split into a part
//       quantised to 2^-11 (all its products with the monomials -- multiples of 1/4 below 13 -- and all their partial sums are
//       EXACT in fp32) and a small remainder: 12 terms, six v_mfma_f32_32x32x2_f32 per 32 splats x 32 pixels.  The result
//       carries the rounding of its last few (tiny) additions only: the same accuracy as the direct d^T Conic d form, and
//       bit-identical between forward and backward (an MFMA output element depends on its operand row and column only), so
//       both passes take identical alpha / saturation decisions.  Saves 9 of the forward's 26 VALU instructions per pair.
//   (2) the backward's wave reductions.  The nine per-(bin, splat) sums (3 colour gradients, six moments of s = dL/dG G about
//       the bin centre, from which dL/dmean2D, dL/dconic, dL/dopacity follow per (bin, splat)) are  SUMS[9, splat] =
//       M[9, pixel] x [w | s][pixel, splat]  with M = {dL/dpixel rgb; 1, u, v, u^2, uv, v^2}: per pair the VALU only forms the
//       two numbers w = alpha T and s, stores them TRANSPOSED through LDS (lane = pixel -> lane = splat column), and sixteen
//       v_mfma_f32_16x16x4_f32 per 8 splats do the 64-lane reductions.  Replaces 8 multiplies + the 22-instruction butterfly
//       reduce-scatter of gsr_composite.hip per (bin, splat).
//
// Everything else (work-ordered XCD-aware bin dispatch, 48-byte record gathers prefetched one round ahead, branch-free blend,
// scalar colour-behind recurrence, atomic-free per-instance gradient records) is as in gsr_composite.hip.
#include "gsr_composite_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr double GSR_LOG2E_D = 1.4426950408889634074;

// hi / lo parts of {c0, cu, cv, cuu, cuv, cvv}, in log2 units (alpha = op * exp2(power2))
struct PowCoef {
    float h[6], l[6];
};

__device__ __forceinline__ void split_hi_lo(double c, float &hi, float &lo) {
    const float cf = (float)c;
    hi = rintf(cf * 2048.f) * (1.f / 2048.f);  // multiple of 2^-11 (|c| < 4096: exact; larger values pass through unchanged)
    lo = (float)(c - (double)hi);
}

// Coefficients of one splat {x, y, conic A B C} about the bin centre (cx, cy):  dx = x - px = X - u,  dy = Y - v
__device__ __forceinline__ PowCoef pow_coefs(float x, float y, float A, float B, float C, float cx, float cy) {
    const double X = (double)x - (double)cx, Y = (double)y - (double)cy;
    const double a = (double)A * GSR_LOG2E_D, b = (double)B * GSR_LOG2E_D, c = (double)C * GSR_LOG2E_D;
    const double cu = a * X + b * Y, cv = b * X + c * Y;
    const double c0 = -0.5 * (X * cu + Y * cv);
    PowCoef p;
    split_hi_lo(c0, p.h[0], p.l[0]);
    split_hi_lo(cu, p.h[1], p.l[1]);
    split_hi_lo(cv, p.h[2], p.l[2]);
    split_hi_lo(-0.5 * a, p.h[3], p.l[3]);
    split_hi_lo(-b, p.h[4], p.l[4]);
    split_hi_lo(-0.5 * c, p.h[5], p.l[5]);
    return p;
}

// monomial m of {1, u, v, u^2, uv, v^2} at pixel index p (0..63) of a bin, (u, v) measured from the bin centre
__device__ __forceinline__ float mono_at(int m, int p) {
    const float u = (float)(p & 7) - 3.5f, v = (float)(p >> 3) - 3.5f;
    return m == 0 ? 1.f : m == 1 ? u : m == 2 ? v : m == 3 ? u * u : m == 4 ? u * v : v * v;
}

// [a_lo | b_lo] , [a_hi | b_hi]  (v_permlane32_swap: the upper half of the first operand trades places with the lower half of the second)
__device__ __forceinline__ void swap32(float a, float b, float &lo_pair, float &hi_pair) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo_pair = __uint_as_float(r[0]);
    hi_pair = __uint_as_float(r[1]);
}

// MFMA operands of the exponent tiles.
//   A (coefficients): v_mfma_f32_32x32x2_f32 wants lane l to hold A[i = l & 31][k = l >> 5].  Lane l staged splat l of the round and
//   holds ITS coefficients; one v_permlane32_swap of (term 2t, term 2t+1) yields the operand of splats 0..31 ([term 2t of splat l |
//   term 2t+1 of splat l-32]) and of splats 32..63 at once.
//   B (monomials): lane l holds B[k = l >> 5][j = l & 31] = monomial (2t + k) % 6 at pixel 32 h + j: constants of the lane.
struct PowOperands {
    float a[2][6];  // [splat half S][k-step t]: t = 0..2 hi terms, 3..5 lo terms
};
__device__ __forceinline__ PowOperands pow_operands(const PowCoef &pc) {
    PowOperands o;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        swap32(pc.h[2 * t], pc.h[2 * t + 1], o.a[0][t], o.a[1][t]);
        swap32(pc.l[2 * t], pc.l[2 * t + 1], o.a[0][3 + t], o.a[1][3 + t]);
    }
    return o;
}
struct PowMono {
    float b[2][3];  // [pixel half h][t % 3]
};
__device__ __forceinline__ PowMono pow_mono(int lane) {
    PowMono m;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int t = 0; t < 3; t++) m.b[h][t] = mono_at(2 * t + (lane >> 5), 32 * h + (lane & 31));
    return m;
}
// exponents of splats 32 S .. 32 S + 31 at pixels 32 h .. 32 h + 31: register r of lane l = splat 32 S + (r & 3) + 8 (r >> 2) + 4 (l >> 5),
// pixel 32 h + (l & 31).  The six hi products and their sums are exact; the lo terms follow.
__device__ __forceinline__ f32x16 pow_tile(const float (&a)[6], const float (&b)[3]) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 6; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t % 3], acc, 0, 0, 0);
    return acc;
}
// the exponents of the 8 staged splats 32 S + 8 q .. + 7 (list order) for THIS lane's pixel, from the two pixel-half tiles of splat half S:
// one v_permlane32_swap per two splats puts pixels 0..31 / 32..63 of one splat side by side
__device__ __forceinline__ void pow_group(const f32x16 &d0, const f32x16 &d1, int q, float (&p)[8]) {
#pragma unroll
    for (int e = 0; e < 4; e++) swap32(d0[4 * q + e], d1[4 * q + e], p[e], p[e + 4]);
}

// Per-lane predicates are kept as wave-uniform 64-bit masks in SGPRs and combined on the scalar unit (one v_cmp per test, never a
// second compare for the complement); __builtin_amdgcn_inverse_ballot_w64 hands a mask back to v_cndmask for free.
typedef unsigned long long lanemask_t;

struct FwdState {
    float T, C0, C1, C2;
    uint32_t last;      // 1-based list position of the last splat that contributed (n_contrib), up to the previous round
    uint32_t last_rnd;  // ... 1-based slot of the last contributor inside the current round (0: none yet): an inline constant per select
    lanemask_t active;  // pixels that are inside the image and not yet saturated
};

// One k-step of the exponent-tile PRODUCTION for a later half round: step M = 0..11 runs through the two pixel-half tiles (6 k-steps
// each) of one splat half.  The production is spread over the blend of the half round before it (three steps per group of 8
// splats), so that every wave's instruction stream is an even mix of VALU and MFMA work: with all waves of a SIMD running the
// same code in near lockstep, long MFMA-only phases would make them queue on the matrix pipe together and then on the VALU
// together (measured: no overlap at all), whereas evenly mixed streams overlap whatever their relative phase.
template <int M>
__device__ __forceinline__ void pow_step(const float (&a)[6], const PowMono &mono, f32x16 &n0, f32x16 &n1) {
    constexpr int h = M / 6, t = M % 6;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 &acc = h ? n1 : n0;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], mono.b[h][t % 3], t == 0 ? zero : acc, 0, 0, 0);
}

template <int S, int Q>
__device__ __forceinline__ void fwd_group(FwdState &st, const f32x16 &d0, const f32x16 &d1, const float4 *__restrict__ wCol, int cnt,
                                          const float (&pa)[6], const PowMono &mono, f32x16 &n0, f32x16 &n1) {
    constexpr int j0 = 32 * S + 8 * Q;
    // groups of 8; between groups one scalar test stops the round as soon as all 64 pixels are saturated
    if (j0 < cnt && st.active != 0ull) {
        float p[8];
        pow_group(d0, d1, Q, p);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int j = j0 + e;
            const float4 c = wCol[j];  // {opacity, r, g, b}: wave-uniform address (LDS broadcast)
            // upstream skips power > 0, which its own (PSD) quadratic form never produces; the split evaluation here can land a few
            // 1e-7 above zero where the true exponent is 0 (splat centre on a pixel centre): clamp instead of skipping
            const float alpha = fminf(0.99f, c.x * __builtin_amdgcn_exp2f(fminf(p[e], 0.f)));
            const lanemask_t valid = st.active & ~__ballot(alpha < 1.f / 255.f);
            const float test_T = __builtin_fmaf(-alpha, st.T, st.T);  // T (1 - alpha)
            const lanemask_t sat = __ballot(test_T < 0.0001f);
            const lanemask_t use_m = valid & ~sat;
            st.active &= ~(valid & sat);
            const bool use = __builtin_amdgcn_inverse_ballot_w64(use_m);
            const float w = use ? alpha * st.T : 0.f;
            st.C0 += c.y * w;
            st.C1 += c.z * w;
            st.C2 += c.w * w;
            st.T = use ? test_T : st.T;
            st.last_rnd = use ? (uint32_t)(j + 1) : st.last_rnd;
            if (e == 1) pow_step<3 * Q>(pa, mono, n0, n1);
            if (e == 4) pow_step<3 * Q + 1>(pa, mono, n0, n1);
            if (e == 7) pow_step<3 * Q + 2>(pa, mono, n0, n1);
        }
    } else {
        pow_step<3 * Q>(pa, mono, n0, n1);
        pow_step<3 * Q + 1>(pa, mono, n0, n1);
        pow_step<3 * Q + 2>(pa, mono, n0, n1);
    }
}

// blend the 32 staged splats of half S from their exponent tiles (d0, d1) while producing the tiles (n0, n1) of the NEXT half from pa
template <int S>
__device__ __forceinline__ void fwd_half(FwdState &st, const f32x16 &d0, const f32x16 &d1, const float4 *__restrict__ wCol, int cnt,
                                         const float (&pa)[6], const PowMono &mono, f32x16 &n0, f32x16 &n1) {
    fwd_group<S, 0>(st, d0, d1, wCol, cnt, pa, mono, n0, n1);
    fwd_group<S, 1>(st, d0, d1, wCol, cnt, pa, mono, n0, n1);
    fwd_group<S, 2>(st, d0, d1, wCol, cnt, pa, mono, n0, n1);
    fwd_group<S, 3>(st, d0, d1, wCol, cnt, pa, mono, n0, n1);
}

__global__ __launch_bounds__(64, 4) void k_composite_fwd_mfma(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                           const uint32_t *__restrict__ bin_offset, const uint32_t *__restrict__ wg_order,
                                                           const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                           float *__restrict__ out_color, float *__restrict__ final_T,
                                                           uint32_t *__restrict__ n_contrib, const GsrHeader *__restrict__ hdr) {
    __shared__ float4 sCol[2][WAVE];  // {opacity, r, g, b} of the 64 staged splats, double buffered (round parity)
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    WaveGeom g = wave_geom(W, H, bx, bin_offset, wg_order, list_pos);
    if (hdr->overflow) {  // nothing can be rendered from truncated lists: a deterministic zero image instead of uninitialised memory
        fwd_write_blank(g, W, H, out_color, final_T, n_contrib);
        return;
    }
    g.r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r0);  // the list range is wave-uniform: keep it (and every loop bound and
    g.r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r1);  // lane mask derived from it) on the scalar unit
    const int lane = g.lane;
    const float cx = (float)(g.px - (lane & 7)) + 3.5f, cy = (float)(g.py - (lane >> 3)) + 3.5f;  // bin centre
    const PowMono mono = pow_mono(lane);

    FwdState st;
    st.T = 1.f; st.C0 = 0.f; st.C1 = 0.f; st.C2 = 0.f;
    st.last = 0; st.last_rnd = 0;
    st.active = __ballot(g.inside);

    // Software pipeline over half rounds (32 staged splats): while half (r, S) is blended, the exponent tiles of the half after it
    // are produced on the matrix pipe; the records of round r + 1 are gathered during round r's first half, turned into
    // coefficients between the halves, and the gather of round r + 2 starts there.
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    // two-stage gather: the list entry (Gaussian id) of a slot is fetched one round before its 48-byte record, so that neither of the
    // two dependent loads ever has less than a whole round to arrive (under load a scattered gather takes microseconds)
    uint32_t nId = 0xffffffffu;
    auto load_id = [&](uint32_t k) { nId = k < g.r1 ? point_list[k] : 0xffffffffu; };
    auto load_rec = [&]() {
        nB.y = 0.f;  // a slot without a splat blends nothing (opacity 0 -> alpha 0 < 1/255; stale x, y, conic stay finite)
        if (nId != 0xffffffffu) {
            const float4 *s = reinterpret_cast<const float4 *>(splats + nId);
            nA = s[0]; nB = s[1]; nC = s[2].x;
        }
    };
    load_id(g.r0 + lane);
    load_rec();                      // round 0
    load_id(g.r0 + WAVE + lane);
    float opS1[6];
    f32x16 tA0, tA1, tB0, tB1;
    {
        const PowOperands op = pow_operands(pow_coefs(nA.x, nA.y, nA.z, nA.w, nB.x, cx, cy));
        sCol[0][lane] = make_float4(nB.y, nB.z, nB.w, nC);
        load_rec();                  // round 1
        load_id(g.r0 + 2 * WAVE + lane);
        tA0 = pow_tile(op.a[0], mono.b[0]);
        tA1 = pow_tile(op.a[0], mono.b[1]);
#pragma unroll
        for (int t = 0; t < 6; t++) opS1[t] = op.a[1][t];
    }
    wave_sync_lds();
    uint32_t par = 0;
    for (uint32_t base = g.r0; base < g.r1; base += WAVE, par ^= 1u) {
        if (st.active == 0ull) break;  // every pixel of this bin is saturated (or outside the image)
        const int cnt = (int)min((uint32_t)WAVE, g.r1 - base);
        const float4 *wCol = sCol[par];
        fwd_half<0>(st, tA0, tA1, wCol, cnt, opS1, mono, tB0, tB1);
        // between the halves: round r + 1's records have arrived
        const PowOperands op = pow_operands(pow_coefs(nA.x, nA.y, nA.z, nA.w, nB.x, cx, cy));
        wave_sync_lds();
        sCol[par ^ 1u][lane] = make_float4(nB.y, nB.z, nB.w, nC);  // the other buffer: last read in the previous round
        wave_sync_lds();
        load_rec();                        // records of round r + 2 (their ids were requested a round ago)
        load_id(base + 3 * WAVE + lane);   // ids of round r + 3
        fwd_half<1>(st, tB0, tB1, wCol, cnt, op.a[0], mono, tA0, tA1);
#pragma unroll
        for (int t = 0; t < 6; t++) opS1[t] = op.a[1][t];
        st.last = st.last_rnd ? (base - g.r0) + st.last_rnd : st.last;
        st.last_rnd = 0;
    }
    if (g.inside) {
        const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
        final_T[q] = st.T;
        n_contrib[q] = st.last;
        out_color[q] = st.C0 + st.T * bg[0];
        out_color[npix + q] = st.C1 + st.T * bg[1];
        out_color[2 * npix + q] = st.C2 + st.T * bg[2];
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// Transposed staging of {w, s} for one group of 8 staged splats: 16 COLUMNS (0..7: w of splat c, 8..15: s of splat c - 8) x 64
// pixels.  Pixel p = 16 k + t lives at word k * 320 + column * 20 + t: a lane's store (ds_write_b32, lane = pixel) is conflict
// free (2-way at most), and the operand fetch (lane (k, column) reads its 16 pixels with four ds_read_b128) is conflict free for
// the 16-lane groups ds_read_b128 is serviced in (5 * column mod 16 is a bijection; 320 = 0 mod 64).
constexpr int XT_KSTRIDE = 320, XT_CSTRIDE = 20, XT_WORDS = 4 * XT_KSTRIDE;

// A operand of the reduction (v_mfma_f32_16x16x4_f32: lane l holds A[i = l & 15][k = l >> 4]): for k-step t the contracted pixel is
// 16 k + t; rows 0..2 = dL/dpixel (r, g, b) of that pixel, rows 3..8 = its monomials, rows 9..15 = 0.
struct RedOperand {
    float a[16];
};
__device__ __forceinline__ RedOperand red_operand(const float *__restrict__ sD /* [3][64] dL/dpixel, LDS */, int lane) {
    const int i = lane & 15, k = lane >> 4;
    const float4 *src = reinterpret_cast<const float4 *>(sD + (i < 3 ? i : 0) * 64 + 16 * k);
    const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    const float dv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    RedOperand o;
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const float m = (i >= 3 && i <= 8) ? mono_at(i - 3, 16 * k + t) : 0.f;
        o.a[t] = i < 3 ? dv[t] : m;
    }
    return o;
}
// D[row = 4 (l >> 4) + r][column = l & 15] = sum over the 64 pixels of A[row][pixel] * X[column][pixel]
struct RedB {
    float v[16];
};
__device__ __forceinline__ RedB red_fetch(const float *__restrict__ sX, int lane) {
    const float4 *src = reinterpret_cast<const float4 *>(sX + (lane >> 4) * XT_KSTRIDE + (lane & 15) * XT_CSTRIDE);
    const float4 b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
    RedB r = {{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w}};
    return r;
}
// k-step TT of the 16: two accumulators (even / odd steps), because back-to-back MFMAs on one accumulator wait 40 instead of 32 cycles
template <int TT>
__device__ __forceinline__ void red_step(const RedOperand &A, const RedB &B, f32x4 &da, f32x4 &db) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 &acc = (TT & 1) ? db : da;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.a[TT], B.v[TT], TT < 2 ? zero : acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 red_group(const RedOperand &A, const float *__restrict__ sX, int lane) {
    const RedB B = red_fetch(sX, lane);
    f32x4 da, db;
    red_step<0>(A, B, da, db); red_step<1>(A, B, da, db); red_step<2>(A, B, da, db); red_step<3>(A, B, da, db);
    red_step<4>(A, B, da, db); red_step<5>(A, B, da, db); red_step<6>(A, B, da, db); red_step<7>(A, B, da, db);
    red_step<8>(A, B, da, db); red_step<9>(A, B, da, db); red_step<10>(A, B, da, db); red_step<11>(A, B, da, db);
    red_step<12>(A, B, da, db); red_step<13>(A, B, da, db); red_step<14>(A, B, da, db); red_step<15>(A, B, da, db);
    return da + db;
}
// Park the sums of the group's 8 splats as 12-float rows {dr, dg, db, m0 | mu, mv, muu, muv | mvv, -, -, -} of sAcc[splat]:
//   lanes 0..7   (column c, rows 0..3):   regs 0..2 = colour sums of splat c          -> words 0..2
//   lanes 24..31 (column 8+c, rows 4..7): regs 0..3 = mu, mv, muu, muv of splat c     -> words 4..7
//   lanes 8..15  (column 8+c, rows 0..3): reg 3 = m0                                  -> word 3
//   lanes 40..47 (column 8+c, rows 8..11): reg 0 = mvv                                -> word 8
__device__ __forceinline__ void red_park(const f32x4 &d, float *__restrict__ accF /* sAcc of staged splat 8 * group */, int lane) {
    // four disjoint sets of lanes and of words: no store of one set may be ordered against another's
    float *row = accF + 12 * (lane & 7);
    if (lane < 8) { row[0] = d[0]; row[1] = d[1]; row[2] = d[2]; }
    if (lane >= 8 && lane < 16) row[3] = d[3];
    if (lane >= 24 && lane < 32) *reinterpret_cast<float4 *>(row + 4) = make_float4(d[0], d[1], d[2], d[3]);
    if (lane >= 40 && lane < 48) row[8] = d[0];
}

struct BwdState {
    float T, A;                    // transmittance in front of the current splat; (colour accumulated behind it) . dL/dpixel
    unsigned long long touched;    // which staged splats of the round received any gradient (wave-uniform)
    int pend;                      // group of 8 whose {w, s} columns sit in sX waiting for their reduction (-1: none), wave-uniform
};

// Group GQ (8 staged splats) of a round: blend back to front and store {w, s} transposed into sX, while the matrix pipe (a) reduces
// the columns the PREVIOUS group left in sX (16 k-steps, two per splat; its operand fetch is issued first, ahead of this group's
// stores: the DS queue of a wave is in order) and (b) produces the exponent tiles of the next half round (3 k-steps).
template <int GQ>
__device__ __forceinline__ void bwd_group(BwdState &st, const f32x16 &d0, const f32x16 &d1, const float4 *__restrict__ wCol, float *__restrict__ sX,
                                          float *__restrict__ accF, const RedOperand &RA, int lane, int xw, uint32_t topu, uint32_t last,
                                          float d_r, float d_g, float d_b, float nTb, const float (&pa)[6], const PowMono &mono, f32x16 &n0,
                                          f32x16 &n1) {
    constexpr int Q = GQ & 3;
    wave_sync_lds();
    const RedB B = red_fetch(sX, lane);
    wave_sync_lds();
    f32x4 da, db;
    float p[8];
    pow_group(d0, d1, Q, p);
    unsigned gt = 0u;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int j = 8 * GQ + e;
        const float4 c = wCol[j];
        const float G = __builtin_amdgcn_exp2f(fminf(p[e], 0.f));  // clamp as in the forward
        const float alpha = fminf(0.99f, c.x * G);
        // staged slot j sits at list position top - j (slots behind the front of the list carry opacity 0: alpha < 1/255)
        const lanemask_t valid_m = __ballot(last > topu - (uint32_t)j) & ~__ballot(alpha < 1.f / 255.f);
        gt |= (valid_m != 0ull) ? (1u << e) : 0u;  // scalar
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
        // Branch-free: a lane this splat does not reach runs the same arithmetic with alpha = 0 and G = 0 (exact no-op), and
        // so does a whole splat nobody reaches (7 % of the walked entries; skipping them individually would put a branch
        // between every two splats and keep the scheduler from fetching the next splat's colour ahead of time).
        // The colour seen behind the splat is carried as its dot product with dL/dpixel: A <- alpha cd + (1 - alpha) A.
        const float Ge = valid ? G : 0.f;
        const float ae = valid ? alpha : 0.f;
        const float om = 1.f - ae;
        const float rcp = __builtin_amdgcn_rcpf(om);
        st.T = st.T * rcp;
        const float cd = c.y * d_r + c.z * d_g + c.w * d_b;
        const float w = ae * st.T;  // dchannel/dcolour
        const float dL_dalpha = (cd - st.A) * st.T + nTb * rcp;
        st.A = ae * cd + om * st.A;
        const float s = (c.x * dL_dalpha) * Ge;  // s = dL/dG * G, dL/dG = opacity * dL/dalpha straight through the 0.99 clamp
        sX[xw + XT_CSTRIDE * e] = w;
        sX[xw + XT_CSTRIDE * (8 + e)] = s;
        if (e == 0) { red_step<0>(RA, B, da, db); red_step<1>(RA, B, da, db); }
        if (e == 1) { red_step<2>(RA, B, da, db); red_step<3>(RA, B, da, db); pow_step<3 * Q>(pa, mono, n0, n1); }
        if (e == 2) { red_step<4>(RA, B, da, db); red_step<5>(RA, B, da, db); }
        if (e == 3) { red_step<6>(RA, B, da, db); red_step<7>(RA, B, da, db); }
        if (e == 4) { red_step<8>(RA, B, da, db); red_step<9>(RA, B, da, db); pow_step<3 * Q + 1>(pa, mono, n0, n1); }
        if (e == 5) { red_step<10>(RA, B, da, db); red_step<11>(RA, B, da, db); }
        if (e == 6) { red_step<12>(RA, B, da, db); red_step<13>(RA, B, da, db); }
        if (e == 7) { red_step<14>(RA, B, da, db); red_step<15>(RA, B, da, db); pow_step<3 * Q + 2>(pa, mono, n0, n1); }
    }
    if (st.pend >= 0) red_park(da + db, accF + 12 * 8 * st.pend, lane);
    st.touched |= (unsigned long long)gt << (8 * GQ);
    st.pend = gt ? GQ : -1;
}

__global__ __launch_bounds__(64, 2) void k_composite_bwd_mfma(int W, int H, int bx, const GsrSplat *__restrict__ splats,
                                                           const uint32_t *__restrict__ bin_offset, const uint32_t *__restrict__ wg_order,
                                                           const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                           const float *__restrict__ dL_dpix, const float *__restrict__ final_T,
                                                           const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ goff,
                                                           const uint32_t *__restrict__ gpart, uint32_t *__restrict__ inst_pos,
                                                           GsrGradAcc *__restrict__ inst_grad, const GsrHeader *__restrict__ hdr) {
    __shared__ float4 sCol[2][WAVE];    // {opacity, r, g, b} of the 64 staged splats, double buffered (round parity)
    __shared__ float4 sAcc[WAVE * 3];   // per staged splat: the nine sums (12 floats)
    __shared__ __attribute__((aligned(16))) float sX[XT_WORDS];
    __shared__ __attribute__((aligned(16))) float sD[3 * WAVE];
    if (hdr->overflow) return;
    const uint32_t list_pos = xcd_list_pos(blockIdx.x, hdr->num_busy_wgs);
    if (list_pos >= hdr->num_busy_wgs) return;  // idle workgroups sit at the end of wg_order
    WaveGeom g = wave_geom(W, H, bx, bin_offset, wg_order, list_pos);
    g.r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r0);
    g.r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.r1);
    if (g.r1 <= g.r0) return;
    const int lane = g.lane;
    const size_t npix = (size_t)W * H, q = (size_t)g.py * W + g.px;
    const float cx = (float)(g.px - (lane & 7)) + 3.5f, cy = (float)(g.py - (lane >> 3)) + 3.5f;  // bin centre
    float *accF = reinterpret_cast<float *>(sAcc);

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

    sD[lane] = d0; sD[WAVE + lane] = d1; sD[2 * WAVE + lane] = d2;
    wave_sync_lds();
    const RedOperand RA = red_operand(sD, lane);
    const PowMono mono = pow_mono(lane);
    const int xw = (lane >> 4) * XT_KSTRIDE + (lane & 15);  // where this lane (= pixel) stores into a column of sX

    BwdState st;
    st.T = T_final; st.A = 0.f; st.touched = 0ull; st.pend = -1;
    const float nTb = -T_final * bg_dot;

    // positions are 0-based from the front of the bin list; walk from max_last-1 down to 0 in rounds of 64.  Software pipeline as in the
    // forward: while a half round is blended, the exponent tiles of the next one are produced; the records of the next round are
    // gathered during the first half and turned into coefficients between the halves.
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
    float nC = 0.f;
    uint32_t nSlot = 0;  // where this lane's staged instance lives in its Gaussian's inst_pos slots
    const int bin_x = g.bin % bx, bin_y = g.bin / bx;
    // two-stage gather as in the forward: list entry one round ahead of the record
    uint32_t nId = 0xffffffffu;
    auto load_id = [&](int64_t pos) { nId = pos >= 0 ? point_list[g.r0 + (uint32_t)pos] : 0xffffffffu; };  // pos < 0: beyond the front of the list
    auto load_rec = [&]() {
        nB.y = 0.f;  // a slot without a splat blends nothing (opacity 0)
        if (nId != 0xffffffffu) {
            const uint32_t id = nId;
            const float4 *sp = reinterpret_cast<const float4 *>(splats + id);
            nA = sp[0]; nB = sp[1];
            const float4 c = sp[2];
            nC = c.x;
            const uint32_t lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
            const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff;
            nSlot = gpart[id >> 10] + goff[id] + (uint32_t)((bin_y - y0) * (x1 - x0) + (bin_x - x0));
        }
    };
    load_id(max_last - 1 - lane);
    load_rec();
    load_id(max_last - 1 - WAVE - lane);
    // records of the round being blended stay in registers for its flush
    float sx = nA.x, sy = nA.y, sA_ = nA.z, sB_ = nA.w, sC_ = nB.x, sop = nB.y;
    uint32_t curSlot = nSlot;
    float opS1[6];
    f32x16 tA0, tA1, tB0, tB1;
    {
        const PowOperands op = pow_operands(pow_coefs(sx, sy, sA_, sB_, sC_, cx, cy));
        sCol[0][lane] = make_float4(sop, nB.z, nB.w, nC);
        load_rec();
        load_id(max_last - 1 - 2 * WAVE - lane);
        tA0 = pow_tile(op.a[0], mono.b[0]);
        tA1 = pow_tile(op.a[0], mono.b[1]);
#pragma unroll
        for (int t = 0; t < 6; t++) opS1[t] = op.a[1][t];
    }
    wave_sync_lds();
    uint32_t par = 0;
    for (int64_t top = max_last - 1; top >= 0; top -= WAVE, par ^= 1u) {
        const int cnt = (int)min((int64_t)WAVE, top + 1);
        const uint32_t topu = (uint32_t)top;  // slot j beyond the front of the list: topu - j wraps to a huge position, never < last
        const float4 *wCol = sCol[par];
        st.touched = 0ull;
        if (0 < cnt) bwd_group<0>(st, tA0, tA1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, opS1, mono, tB0, tB1);
        if (8 < cnt) bwd_group<1>(st, tA0, tA1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, opS1, mono, tB0, tB1);
        if (16 < cnt) bwd_group<2>(st, tA0, tA1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, opS1, mono, tB0, tB1);
        if (24 < cnt) bwd_group<3>(st, tA0, tA1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, opS1, mono, tB0, tB1);
        // between the halves: the next round's records have arrived (only a full round has a successor: cnt < 64 is the front of the list)
        const float nx = nA.x, ny = nA.y, nAa = nA.z, nAb = nA.w, nCc = nB.x, nop = nB.y;
        const uint32_t nSl = nSlot;
        const PowOperands op = pow_operands(pow_coefs(nx, ny, nAa, nAb, nCc, cx, cy));
        wave_sync_lds();
        sCol[par ^ 1u][lane] = make_float4(nop, nB.z, nB.w, nC);  // the other buffer: last read in the previous round
        wave_sync_lds();
        load_rec();                        // records two rounds ahead (their ids were requested a round ago)
        load_id(top - 3 * WAVE - lane);
        if (32 < cnt) bwd_group<4>(st, tB0, tB1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, op.a[0], mono, tA0, tA1);
        if (40 < cnt) bwd_group<5>(st, tB0, tB1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, op.a[0], mono, tA0, tA1);
        if (48 < cnt) bwd_group<6>(st, tB0, tB1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, op.a[0], mono, tA0, tA1);
        if (56 < cnt) bwd_group<7>(st, tB0, tB1, wCol, sX, accF, RA, lane, xw, topu, last, d0, d1, d2, nTb, op.a[0], mono, tA0, tA1);
#pragma unroll
        for (int t = 0; t < 6; t++) opS1[t] = op.a[1][t];
        if (st.pend >= 0) {  // the round's last group: reduce it now (the flush below needs it)
            wave_sync_lds();
            red_park(red_group(RA, sX, lane), accF + 12 * 8 * st.pend, lane);
            st.pend = -1;
        }
        const unsigned long long touched = st.touched;
        wave_sync_lds();
        if ((touched >> lane) & 1ull) {  // lane j turns staged splat j's sums into ONE 48-byte instance record (no atomics)
            const float4 v0 = sAcc[3 * lane], v1 = sAcc[3 * lane + 1], v2 = sAcc[3 * lane + 2];
            // moments about the bin centre -> moments about the splat centre (dx = X - u, dy = Y - v)
            const float X = sx - cx, Y = sy - cy;
            const float m0 = v0.w, mu = v1.x, mv = v1.y, muu = v1.z, muv = v1.w, mvv = v2.x;
            const float Sx = X * m0 - mu, Sy = Y * m0 - mv;
            const float Sxx = X * (Sx - mu) + muu;
            const float Sxy = X * Sy - Y * mu + muv;
            const float Syy = Y * (Sy - mv) + mvv;
            // dG/d(delta) = -G (A dx + B dy), -G (C dy + B dx);  dL/dconic = -0.5 s {dx^2, dx dy, dy^2};  dL/dop = G dL/dalpha = s / op
            const float g_mx = ddelx_dx * (-sA_ * Sx - sB_ * Sy);
            const float g_my = ddely_dy * (-sC_ * Sy - sB_ * Sx);
            const uint32_t pp = g.r0 + (uint32_t)(top - lane);  // consecutive lanes -> consecutive records: coalesced
            float4 *dst = reinterpret_cast<float4 *>(inst_grad + pp);
            dst[0] = make_float4(v0.x, v0.y, v0.z, g_mx);
            dst[1] = make_float4(g_my, -0.5f * Sxx, -0.5f * Sxy, -0.5f * Syy);
            dst[2] = make_float4(m0 * __builtin_amdgcn_rcpf(sop), 0.f, 0.f, 0.f);
            inst_pos[curSlot] = pp;
        }
        sx = nx; sy = ny; sA_ = nAa; sB_ = nAb; sC_ = nCc; sop = nop; curSlot = nSl;
    }
}

// ---- device self-test of the matrix-core plumbing (gsr_selftest): the SAME device functions the kernels use, on pseudo-random
// operands, against straightforward per-lane evaluation.  out[0] = max |exponent tile - fp64 polynomial| / (1 + |value|),
// out[1] = max relative error of the nine reduction sums, out[2] = 1 if v_permlane32_swap behaves as documented.
__global__ __launch_bounds__(64) void k_selftest(float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sX[XT_WORDS];
    __shared__ __attribute__((aligned(16))) float sD[3 * WAVE];
    __shared__ float4 sAcc[WAVE * 3];
    __shared__ float sP[WAVE][WAVE + 1];  // [splat][pixel] exponents
    __shared__ float sRec[WAVE][5];
    const int lane = threadIdx.x;
    auto rnd = [](uint32_t a) {
        a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
        return (float)(a & 0xffffff) * (1.f / 16777216.f);
    };
    // (0) swap semantics
    float lo, hi;
    swap32((float)lane, (float)(lane + 100), lo, hi);
    const bool swap_ok = (lane < 32 ? (lo == (float)lane && hi == (float)(lane + 32)) : (lo == (float)(lane + 100 - 32) && hi == (float)(lane + 100)));
    // (1) exponent tiles: lane = splat for the coefficients, lane = pixel for the result
    const float cx = 515.5f, cy = 259.5f;
    const float x = cx + (rnd(lane * 7 + 1) - 0.5f) * 30.f, y = cy + (rnd(lane * 7 + 2) - 0.5f) * 30.f;
    const float a = 0.05f + 3.f * rnd(lane * 7 + 3), c = 0.05f + 3.f * rnd(lane * 7 + 4), b = (rnd(lane * 7 + 5) - 0.5f) * 1.8f * sqrtf(a * c);
    sRec[lane][0] = x; sRec[lane][1] = y; sRec[lane][2] = a; sRec[lane][3] = b; sRec[lane][4] = c;
    const PowOperands op = pow_operands(pow_coefs(x, y, a, b, c, cx, cy));
    const PowMono mono = pow_mono(lane);
    f32x16 dt[2][2];
    dt[0][0] = pow_tile(op.a[0], mono.b[0]); dt[0][1] = pow_tile(op.a[0], mono.b[1]);
    dt[1][0] = pow_tile(op.a[1], mono.b[0]); dt[1][1] = pow_tile(op.a[1], mono.b[1]);
#pragma unroll
    for (int gq = 0; gq < 8; gq++) {
        float p[8];
        pow_group(dt[gq >> 2][0], dt[gq >> 2][1], gq & 3, p);
#pragma unroll
        for (int e = 0; e < 8; e++) sP[8 * gq + e][lane] = p[e];
    }
    __syncthreads();
    float e0 = 0.f;
    {
        const double px = (double)cx - 3.5 + (lane & 7), py = (double)cy - 3.5 + (lane >> 3);
        for (int j = 0; j < WAVE; j++) {
            const double dx = (double)sRec[j][0] - px, dy = (double)sRec[j][1] - py;
            const double ref = (-0.5 * ((double)sRec[j][2] * dx * dx + (double)sRec[j][4] * dy * dy) - (double)sRec[j][3] * dx * dy) * GSR_LOG2E_D;
            e0 = fmaxf(e0, (float)(fabs((double)sP[j][lane] - ref) / (1.0 + fabs(ref))));
        }
    }
    // (2) reductions: 8 splats, lane = pixel
    const float d0 = rnd(lane * 5 + 1000) - 0.5f, d1 = rnd(lane * 5 + 1001) - 0.5f, d2 = rnd(lane * 5 + 1002) - 0.5f;
    sD[lane] = d0; sD[WAVE + lane] = d1; sD[2 * WAVE + lane] = d2;
    __syncthreads();
    const RedOperand RA = red_operand(sD, lane);
    const int xw = (lane >> 4) * XT_KSTRIDE + (lane & 15);
    float wv[8], sv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        wv[e] = rnd(lane * 16 + e + 5000);
        sv[e] = rnd(lane * 16 + 8 + e + 5000) - 0.5f;
        sX[xw + XT_CSTRIDE * e] = wv[e];
        sX[xw + XT_CSTRIDE * (8 + e)] = sv[e];
    }
    __syncthreads();
    const f32x4 d = red_group(RA, sX, lane);
    red_park(d, reinterpret_cast<float *>(sAcc), lane);
    __syncthreads();
    float e1 = 0.f;
    {
        const float u = (float)(lane & 7) - 3.5f, v = (float)(lane >> 3) - 3.5f;
        const float *accF = reinterpret_cast<const float *>(sAcc);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float t[9] = {wv[e] * d0, wv[e] * d1, wv[e] * d2, sv[e], sv[e] * u, sv[e] * v, sv[e] * u * u, sv[e] * u * v, sv[e] * v * v};
            float ta[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                float r = t[k], ra = fabsf(t[k]);
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) { r += __shfl_xor(r, dd, 64); ra += __shfl_xor(ra, dd, 64); }
                t[k] = r; ta[k] = ra;
            }
#pragma unroll
            for (int k = 0; k < 9; k++) e1 = fmaxf(e1, fabsf(accF[12 * e + k] - t[k]) / (1e-6f + ta[k]));
        }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { e0 = fmaxf(e0, __shfl_xor(e0, dd, 64)); e1 = fmaxf(e1, __shfl_xor(e1, dd, 64)); }
    const unsigned long long okm = __ballot(swap_ok);
    if (lane == 0) { out[0] = e0; out[1] = e1; out[2] = (okm == ~0ull) ? 1.f : 0.f; out[3] = 0.f; }
}

}  // namespace

void gsr_launch_composite_fwd_mfma(int W, int H, int bx, int by, const GsrSplat *splats, const uint32_t *bin_offset, const uint32_t *wg_order,
                                   const uint32_t *point_list, const float *bg, float *out_color, float *final_T, uint32_t *n_contrib,
                                   const GsrHeader *hdr, hipStream_t s) {
    const int wgs = bx * by;
    if (wgs <= 0) return;
    hipLaunchKernelGGL(k_composite_fwd_mfma, dim3(wgs), dim3(64), gsr_debug_lds_pad(), s, W, H, bx, splats, bin_offset, wg_order, point_list, bg, out_color,
                       final_T, n_contrib, hdr);
}

void gsr_launch_composite_bwd_mfma(int W, int H, int bx, int by, const GsrSplat *splats, const uint32_t *bin_offset, const uint32_t *wg_order,
                                   const uint32_t *point_list, const float *bg, const float *dL_dpix, const float *final_T,
                                   const uint32_t *n_contrib, const uint32_t *goff, const uint32_t *gpart, uint32_t *inst_pos,
                                   GsrGradAcc *inst_grad, const GsrHeader *hdr, hipStream_t s) {
    const int wgs = bx * by;
    if (wgs <= 0) return;
    hipLaunchKernelGGL(k_composite_bwd_mfma, dim3(wgs), dim3(64), gsr_debug_lds_pad(), s, W, H, bx, splats, bin_offset, wg_order, point_list, bg, dL_dpix,
                       final_T, n_contrib, goff, gpart, inst_pos, inst_grad, hdr);
}

void gsr_launch_selftest(float *out, hipStream_t s) { hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, s, out); }
