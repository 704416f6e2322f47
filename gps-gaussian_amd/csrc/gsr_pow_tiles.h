// gsr_pow_tiles.h -- the exponents of 64 staged splats x 64 pixels of a bin on the MATRIX CORES (bf16 MFMA), exactly.
//
// power(pixel, splat) = -1/2 d^T Conic d is a quadratic polynomial in the pixel coordinates (u, v) measured from the bin centre:
//     power = c0 + cu u + cv v + cuu u^2 + cuv u v + cvv v^2,        i.e.   POWER[splat, pixel] = COEF[splat, 6] x MONO[6, pixel].
// Evaluating it per (pixel, splat) pair costs the vector ALUs 8 instructions (2 subtractions + the 6-instruction form of
// gsr_power2); as a matrix product it costs them nothing per pair, and on gfx950 the bf16 MFMA is the only matrix instruction that
// is cheap in SIMD time (v_mfma_f32_32x32x16_bf16: 32 cycles for 16,384 multiply-adds; the f32-input MFMAs run at the vector rate
// and do NOT overlap with VALU work -- measured, tools/ubench/mfma_coexec.hip, SQ_VALU_MFMA_COEXEC_CYCLES = 0 -- so they only move
// the cost around).  bf16 has 8 significant bits, and the expansion cancels (|c0| can be 100 where the value is -5); both are
// handled by making the arithmetic EXACT instead of approximate:
//   * every coefficient is formed in fp64 once per (bin, splat), in log2 units, and split into a part on the 2^-11 grid ("hi") and
//     a remainder |lo| <= 2^-12;
//   * hi is cut into bf16 pieces by truncation (3 pieces for c0, cu, cv, 2 for the quadratic coefficients whose magnitude is
//     bounded by the +0.3 dilation of the 2D covariance): each piece is a multiple of 2^-11 with <= 8 significant bits; the
//     monomials {1, u, v, u^2, uv, v^2} at pixel centres are multiples of 1/4 below 13 (<= 6 bits): every product and every partial
//     sum of the 15 hi terms is exactly representable in the fp32 accumulator (grid 2^-13, magnitude < 2^11): MFMA #0 (K = 16)
//     returns the exact hi part of the exponent, whatever order the hardware adds in;
//   * lo is cut into 2 bf16 pieces (relative error 2^-16 of a number below 2^-12); MFMA #1 adds the 12 lo terms to the (small)
//     exact sum: the result carries one fp32 rounding of the final value, i.e. it is at least as accurate as the direct form.
// Forward and backward build the same tiles from the same records, so they take bit-identical alpha decisions.
//
// Layouts (v_mfma_f32_32x32x16_bf16): A: lane l holds A[i = l & 31][k = 8 (l >> 5) + 0..7] (8 bf16 in 4 VGPRs), B: lane l holds
// B[k = 8 (l >> 5) + 0..7][j = l & 31]; C/D: register r of lane l = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].  Here i = splat
// (32 per tile), j = pixel (32 per tile), k = term.  Lane l stages splat l of the round and computes ITS terms; v_permlane32_swap of
// the packed dwords for k < 8 with those for k >= 8 yields the A operands of splats 0..31 and 32..63 at once.
#pragma once
#include "gsr_composite_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short gsr_bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr double GSR_LOG2E_D = 1.4426950408889634074;

// [a_lo | b_lo] , [a_hi | b_hi]  (v_permlane32_swap: the upper half of the first operand trades places with the lower half of the second)
__device__ __forceinline__ void gsr_swap32(uint32_t a, uint32_t b, uint32_t &lo_pair, uint32_t &hi_pair) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    lo_pair = r[0];
    hi_pair = r[1];
}
// log2(opacity) for pow_terms (v_log_f32: ~1 ulp, i.e. <= 5e-7 absolute for opacities >= 1/255); a slot without a splat (opacity 0) gets an
// exponent that blends nothing
__device__ __forceinline__ float gsr_log2_opacity(float op) { return op > 0.f ? __builtin_amdgcn_logf(op) : -150.f; }
__device__ __forceinline__ float gsr_bf16_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
// two floats -> {bf16(first) in the low half, bf16(second) in the high half}, by truncation (v_perm_b32)
__device__ __forceinline__ uint32_t gsr_pk_bf16(float first, float second) {
    return __builtin_amdgcn_perm(__float_as_uint(second), __float_as_uint(first), 0x07060302u);
}

// the 32 bf16 terms of one splat, packed: t[m][q] = terms k = 2q, 2q+1 of MFMA m (m = 0: hi pieces, m = 1: lo pieces)
struct PowTerms {
    uint32_t t[2][8];
};
// k -> term:  MFMA 0: c0 c0 c0 cu cu cu cv cv | cv cuu cuu cuv cuv cvv cvv 0      MFMA 1: c0 c0 cu cu cv cv cuu cuu | cuv cuv cvv cvv 0 0 0 0
// lop = log2(opacity) rides in the constant term: the tile then yields log2(opacity * G) and the blend loops start at alpha = min(0.99, exp2(.))
__device__ __forceinline__ PowTerms pow_terms(float x, float y, float A, float B, float C, float cx, float cy, float lop) {
    const double X = (double)x - (double)cx, Y = (double)y - (double)cy;  // dx = x - px = X - u,  dy = Y - v
    const double a = (double)A * GSR_LOG2E_D, b = (double)B * GSR_LOG2E_D, c = (double)C * GSR_LOG2E_D;
    const double cu = a * X + b * Y, cv = b * X + c * Y;
    const double coef[6] = {(double)lop - 0.5 * (X * cu + Y * cv), cu, cv, -0.5 * a, -b, -0.5 * c};
    float p1[6], p2[6], p3[6], l1[6], l2[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float cf = (float)coef[i];
        const float hi = rintf(cf * 2048.f) * (1.f / 2048.f);  // multiple of 2^-11 (|c| < 4096: exact)
        const float lo = (float)(coef[i] - (double)hi);
        p1[i] = gsr_bf16_trunc(hi);
        const float r = hi - p1[i];  // exact
        p2[i] = gsr_bf16_trunc(r);
        p3[i] = r - p2[i];           // exact; <= 3 significant bits while |c0| < 256, |cu|, |cv| < 64 (beyond: truncated, the pair is far below 1/255 anyway)
        l1[i] = gsr_bf16_trunc(lo);
        l2[i] = lo - l1[i];
    }
    PowTerms o;
    o.t[0][0] = gsr_pk_bf16(p1[0], p2[0]); o.t[0][1] = gsr_pk_bf16(p3[0], p1[1]); o.t[0][2] = gsr_pk_bf16(p2[1], p3[1]); o.t[0][3] = gsr_pk_bf16(p1[2], p2[2]);
    o.t[0][4] = gsr_pk_bf16(p3[2], p1[3]); o.t[0][5] = gsr_pk_bf16(p2[3], p1[4]); o.t[0][6] = gsr_pk_bf16(p2[4], p1[5]); o.t[0][7] = gsr_pk_bf16(p2[5], 0.f);
    o.t[1][0] = gsr_pk_bf16(l1[0], l2[0]); o.t[1][1] = gsr_pk_bf16(l1[1], l2[1]); o.t[1][2] = gsr_pk_bf16(l1[2], l2[2]); o.t[1][3] = gsr_pk_bf16(l1[3], l2[3]);
    o.t[1][4] = gsr_pk_bf16(l1[4], l2[4]); o.t[1][5] = gsr_pk_bf16(l1[5], l2[5]); o.t[1][6] = 0u; o.t[1][7] = 0u;
    return o;
}
// the monomial each term multiplies (index into {1, u, v, u^2, uv, v^2}; 6 = zero padding)
__device__ __forceinline__ int pow_mono_index(int m, int k) {
    constexpr int M0[16] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5, 6};
    constexpr int M1[16] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 6, 6};
    return m == 0 ? M0[k] : M1[k];
}
__device__ __forceinline__ float pow_mono_value(int idx, int pixel) {
    const float u = (float)(pixel & 7) - 3.5f, v = (float)(pixel >> 3) - 3.5f;
    return idx == 0 ? 1.f : idx == 1 ? u : idx == 2 ? v : idx == 3 ? u * u : idx == 4 ? u * v : idx == 5 ? v * v : 0.f;
}

struct PowOperandsA {
    uint32_t a[2][2][4];  // [splat half S][MFMA m][dword]
};
__device__ __forceinline__ PowOperandsA pow_operands_a(const PowTerms &t) {
    PowOperandsA o;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) gsr_swap32(t.t[m][q], t.t[m][4 + q], o.a[0][m][q], o.a[1][m][q]);
    return o;
}
struct PowOperandsB {
    uint32_t b[2][2][4];  // [pixel half h][MFMA m][dword]: lane l: terms k = 8 (l >> 5) + 2 q, + 1 at pixel 32 h + (l & 31)
};
__device__ __forceinline__ PowOperandsB pow_operands_b(int lane) {
    PowOperandsB o;
    const int kb = lane >> 5;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int pixel = 32 * h + (lane & 31);
                const float lo0 = pow_mono_value(pow_mono_index(m, 2 * q), pixel), lo1 = pow_mono_value(pow_mono_index(m, 8 + 2 * q), pixel);
                const float hi0 = pow_mono_value(pow_mono_index(m, 2 * q + 1), pixel), hi1 = pow_mono_value(pow_mono_index(m, 8 + 2 * q + 1), pixel);
                o.b[h][m][q] = gsr_pk_bf16(kb ? lo1 : lo0, kb ? hi1 : hi0);
            }
    return o;
}
__device__ __forceinline__ gsr_bf16x8 gsr_as_bf16x8(const uint32_t (&d)[4]) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {d[0], d[1], d[2], d[3]};
    return __builtin_bit_cast(gsr_bf16x8, v);
}
// exponents (log2 units) of splats 32 S .. 32 S + 31 at pixels 32 h .. 32 h + 31: register r of lane l = splat 32 S + (r & 3) + 8 (r >> 2)
// + 4 (l >> 5), pixel 32 h + (l & 31)
__device__ __forceinline__ f32x16 pow_tile_bf16(const uint32_t (&a)[2][4], const uint32_t (&b)[2][4]) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gsr_as_bf16x8(a[0]), gsr_as_bf16x8(b[0]), acc, 0, 0, 0);  // hi pieces: exact
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gsr_as_bf16x8(a[1]), gsr_as_bf16x8(b[1]), acc, 0, 0, 0);  // + lo pieces
    return acc;
}
// the exponents of the 8 staged splats 32 S + 8 q .. + 7 (slot order) for THIS lane's pixel, from the two pixel-half tiles of splat half S:
// one v_permlane32_swap per two splats puts pixels 0..31 / 32..63 of one splat side by side
__device__ __forceinline__ void pow_group8(const f32x16 &d0, const f32x16 &d1, int q, float (&p)[8]) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        uint32_t lo, hi;
        gsr_swap32(__float_as_uint(d0[4 * q + e]), __float_as_uint(d1[4 * q + e]), lo, hi);
        p[e] = __uint_as_float(lo);
        p[e + 4] = __uint_as_float(hi);
    }
}

}  // namespace
