// corr_pyramid.hip -- the neighbours of the 1-D correlation sampler inside RAFT-Stereo, for gfx950:
//
//   cv_build_forward / cv_build_backward     all-pairs 1-D correlation volume + its average-pool pyramid
//                                            (/root/reference/core/corr.py:31-61: CorrBlockFast1D.__init__ / .corr)
//   cs_lookup_forward / cs_lookup_backward   the 2r+1-tap lookup in ALL pyramid levels in one launch, written straight into the
//                                            concatenated [N, L*(2r+1), H, W] tensor (core/corr.py:44-51: CorrBlockFast1D.__call__)
//   cu_upsample_forward / cu_upsample_backward   convex-combination upsampling of the flow (core/raft_stereo_human.py:69-81)
//
// The volume is GEMM-shaped work -- per (image, row): C[w1, w2] = sum_d F1[d, w1] F2[d, w2] / sqrt(D), 128 x 192 x 128 at the
// reference's sizes, 2B * 128 of them -- so it runs on the matrix cores: fp32 MFMA (v_mfma_f32_32x32x2_f32), one 32 x 32
// accumulator tile per wave64, operands staged through LDS in K-major order so every MFMA operand fetch is a conflict-free
// 128-byte row.  The epilogue applies 1/sqrt(D) and produces the whole pyramid from the accumulators (neighbouring w2
// columns live in neighbouring lanes: three lane-xor exchanges), so the level-0 volume is never re-read.  The backward
// folds the four per-level gradients back into d(level 0) while it stages that operand, and runs the two transposed GEMMs
// with the same tile engine.  fp16 tensors (stage 2 runs under AMP) are widened on load and rounded once on store.
#include <hip/hip_fp16.h>

#include "gsr_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<__half>(const __half *p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<__half>(__half *p, float v) { *p = __float2half(v); }
template <typename T> __device__ __forceinline__ float rnd(float v);  // round to the storage type (fp16 pyramids are pooled level by level)
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<__half>(float v) { return __half2float(__float2half(v)); }

constexpr int TK = 32;  // K-chunk staged through LDS per step

// ---- tile engine ------------------------------------------------------------------------------------------------------
// C[m, n] = sum_k A(m, k) B(k, n) for one workgroup tile of (64 MT) x (64 NT): 4 waves in a 2 x 2 arrangement, each holding
// MT x NT accumulator tiles of 32 x 32 (rows wm + 64 i, columns wn + 64 j).  fetchA(m, k) / fetchB(k, n) return FOUR operand
// elements (0 outside the problem) starting at that position and running along the operand's contiguous memory index --
// k if A_K_CONTIG / B_K_CONTIG, else m / n -- so the staging loads are 16-byte (fp32) or 8-byte (fp16) vectors that coalesce.
// The next K-chunk is fetched into registers while the MFMAs of the current one run (one LDS buffer, two barriers per
// chunk).  Register v of lane l of an accumulator is row (v / 4) * 8 + (l / 32) * 4 + v % 4, column l % 32.
template <int MT, int NT, bool A_K_CONTIG, bool B_K_CONTIG, typename FA, typename FB>
__device__ __forceinline__ void tile_gemm(int K, FA fetchA, FB fetchB, float (*As)[64 * MT + 1], float (*Bs)[64 * NT + 1], v16f (&acc)[MT][NT]) {
    constexpr int AM = 64 * MT, BN = 64 * NT, NA = TK * AM / 1024, NBL = TK * BN / 1024;  // float4 groups per thread
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = (wid & 1) * 32, wn = (wid >> 1) * 32;
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[i][j][v] = 0.f;
    float4 ra[NA], rb[NBL];
    auto a_idx = [&](int u, int &kk, int &mm) {  // first element of group u of this thread
        const int e = tid + u * 256;
        if (A_K_CONTIG) { kk = (e % (TK / 4)) * 4; mm = e / (TK / 4); } else { mm = (e % (AM / 4)) * 4; kk = e / (AM / 4); }
    };
    auto b_idx = [&](int u, int &kk, int &nn) {
        const int e = tid + u * 256;
        if (B_K_CONTIG) { kk = (e % (TK / 4)) * 4; nn = e / (TK / 4); } else { nn = (e % (BN / 4)) * 4; kk = e / (BN / 4); }
    };
    auto load = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NA; u++) {
            int kk, mm;
            a_idx(u, kk, mm);
            ra[u] = fetchA(mm, k0 + kk);
        }
#pragma unroll
        for (int u = 0; u < NBL; u++) {
            int kk, nn;
            b_idx(u, kk, nn);
            rb[u] = fetchB(k0 + kk, nn);
        }
    };
    load(0);
    for (int k0 = 0; k0 < K; k0 += TK) {
        __syncthreads();  // every wave is done reading the previous chunk
#pragma unroll
        for (int u = 0; u < NA; u++) {
            int kk, mm;
            a_idx(u, kk, mm);
            if (A_K_CONTIG) { As[kk][mm] = ra[u].x; As[kk + 1][mm] = ra[u].y; As[kk + 2][mm] = ra[u].z; As[kk + 3][mm] = ra[u].w; }
            else { As[kk][mm] = ra[u].x; As[kk][mm + 1] = ra[u].y; As[kk][mm + 2] = ra[u].z; As[kk][mm + 3] = ra[u].w; }
        }
#pragma unroll
        for (int u = 0; u < NBL; u++) {
            int kk, nn;
            b_idx(u, kk, nn);
            if (B_K_CONTIG) { Bs[kk][nn] = rb[u].x; Bs[kk + 1][nn] = rb[u].y; Bs[kk + 2][nn] = rb[u].z; Bs[kk + 3][nn] = rb[u].w; }
            else { Bs[kk][nn] = rb[u].x; Bs[kk][nn + 1] = rb[u].y; Bs[kk][nn + 2] = rb[u].z; Bs[kk][nn + 3] = rb[u].w; }
        }
        __syncthreads();
        if (k0 + TK < K) load(k0 + TK);  // in flight while the matrix cores work on this chunk
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) a[i] = As[kk + (lane >> 5)][wm + 64 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; j++) b[j] = Bs[kk + (lane >> 5)][wn + 64 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

// four consecutive elements p[0..3] of which the first `n` (0..4) exist; `vec` = the 4-element vector load is aligned and allowed
template <typename T> __device__ __forceinline__ float4 ld4(const T *p, int n, bool vec);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p, int n, bool vec) {
    if (vec && n >= 4) return *reinterpret_cast<const float4 *>(p);
    return make_float4(n > 0 ? p[0] : 0.f, n > 1 ? p[1] : 0.f, n > 2 ? p[2] : 0.f, n > 3 ? p[3] : 0.f);
}
template <> __device__ __forceinline__ float4 ld4<__half>(const __half *p, int n, bool vec) {
    if (vec && n >= 4) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        const float2 lo = __half22float2(*reinterpret_cast<const __half2 *>(&r.x)), hi = __half22float2(*reinterpret_cast<const __half2 *>(&r.y));
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    return make_float4(n > 0 ? __half2float(p[0]) : 0.f, n > 1 ? __half2float(p[1]) : 0.f, n > 2 ? __half2float(p[2]) : 0.f,
                       n > 3 ? __half2float(p[3]) : 0.f);
}
template <typename T> __device__ __forceinline__ bool vec4_ok(const T *base, int row_len) {  // every row start and every 4-group aligned
    return (row_len & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & (4 * sizeof(T) - 1)) == 0;
}

// ---- forward: volume + pyramid ----------------------------------------------------------------------------------------
struct PyrPtrs {
    void *p[4];
};
struct PyrConstPtrs {
    const void *p[4];
};

constexpr int FWD_MT = 2;  // 128 w1 rows per workgroup: at the reference's 128-wide feature maps F2 is fetched once per (image, row)

template <typename T>
__global__ __launch_bounds__(256) void k_cv_fwd(const T *__restrict__ f1, const T *__restrict__ f2, PyrPtrs pyr, int D, int H, int W1, int W2,
                                                int levels, float scale) {
    __shared__ float As[TK][64 * FWD_MT + 1];
    __shared__ float Bs[TK][64 + 1];
    const int g = blockIdx.z;  // n * H + h
    const int n = g / H, h = g - n * H;
    const int m0 = blockIdx.y * 64 * FWD_MT, n0 = blockIdx.x * 64;
    const T *a = f1 + ((size_t)n * D * H + h) * W1;  // element (d, w1) at a[d * H * W1 + w1]
    const T *b = f2 + ((size_t)n * D * H + h) * W2;
    const size_t lda = (size_t)H * W1, ldb = (size_t)H * W2;
    v16f acc[FWD_MT][1];
    const bool va = vec4_ok(f1, W1), vb = vec4_ok(f2, W2);
    tile_gemm<FWD_MT, 1, false, false>(
        D, [&](int mm, int k) { return ld4(a + (size_t)k * lda + m0 + mm, k < D ? W1 - (m0 + mm) : 0, va); },
        [&](int k, int nn) { return ld4(b + (size_t)k * ldb + n0 + nn, k < D ? W2 - (n0 + nn) : 0, vb); }, As, Bs, acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = n0 + (wid >> 1) * 32 + (lane & 31);  // w2
    // widths of the pyramid levels: avg_pool2d([1, 2], stride [1, 2]) floors
    int wl[4];
    wl[0] = W2;
#pragma unroll
    for (int l = 1; l < 4; l++) wl[l] = wl[l - 1] >> 1;
#pragma unroll
    for (int i = 0; i < FWD_MT; i++) {
        const int row0 = m0 + 64 * i + (wid & 1) * 32 + (lane >> 5) * 4;
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int w1 = row0 + (v >> 2) * 8 + (v & 3);
            float c = rnd<T>(acc[i][0][v] * scale);
            const bool row_ok = w1 < W1;
            const size_t rowbase = (size_t)g * W1 + w1;
            if (row_ok && col < W2) stf(reinterpret_cast<T *>(pyr.p[0]) + rowbase * wl[0] + col, c);
            // level l+1 = 0.5 (even + odd neighbour) of level l: the neighbour sits 2^l lanes away
#pragma unroll
            for (int l = 1; l < 4; l++) {
                const float o = __shfl_xor(c, 1 << (l - 1), 64);
                c = rnd<T>(0.5f * (c + o));
                if (l < levels) {
                    const int cl = col >> l;
                    if (row_ok && (col & ((1 << l) - 1)) == 0 && cl < wl[l]) stf(reinterpret_cast<T *>(pyr.p[l]) + rowbase * wl[l] + cl, c);
                }
            }
        }
    }
}

// ---- backward: d(level 0) folded from the per-level gradients, then the two transposed GEMMs ------------------------------
// d(level 0)[w1, w2] = g0 + g1[w2 / 2] / 2 + g2[w2 / 4] / 4 + g3[w2 / 8] / 8, each term only where the pooling window is complete
template <typename T>
__device__ __forceinline__ float fold_grad(const PyrConstPtrs &gp, int levels, size_t rowbase, int w2, int W2) {
    float s = 0.f, wgt = 1.f;
    int wl = W2, c = w2;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        if (l < levels && gp.p[l] && c < wl) s += wgt * ldf(reinterpret_cast<const T *>(gp.p[l]) + rowbase * wl + c);
        wl >>= 1;
        c >>= 1;
        wgt *= 0.5f;
    }
    return s;
}

constexpr int BWD_MT = 3;  // 192 feature channels per workgroup: the folded gradient volume (the big operand) is fetched once per GEMM

// grad_f1[d, w1] = scale * sum_w2 F2[d, w2] dC[w1, w2]      (M = d, N = w1, K = w2; both operands contiguous in k)
// grad_f2[d, w2] = scale * sum_w1 F1[d, w1] dC[w1, w2]      (M = d, N = w2, K = w1; A contiguous in k, B contiguous in n)
template <typename T, bool WRT_F1>
__global__ __launch_bounds__(256, 2) void k_cv_bwd(const T *__restrict__ fother, PyrConstPtrs gp, T *__restrict__ gout, int D, int H, int W1, int W2,
                                                int levels, float scale) {
    __shared__ float As[TK][64 * BWD_MT + 1];
    __shared__ float Bs[TK][64 + 1];
    const int g = blockIdx.z;
    const int n = g / H, h = g - n * H;
    const int m0 = blockIdx.y * 64 * BWD_MT, n0 = blockIdx.x * 64;  // m = d, n = w1 (WRT_F1) or w2
    const int Wk = WRT_F1 ? W2 : W1, Wn = WRT_F1 ? W1 : W2;
    const T *fo = fother + ((size_t)n * D * H + h) * Wk;  // (d, k) at fo[d * H * Wk + k]
    const size_t ldf_ = (size_t)H * Wk;
    v16f acc[BWD_MT][1];
    const bool vf = vec4_ok(fother, Wk);
    auto fa = [&](int mm, int k) { return ld4(fo + (size_t)(m0 + mm) * ldf_ + k, m0 + mm < D ? Wk - k : 0, vf); };  // 4 along k
    // four consecutive w2 of one volume row, folded from the per-level gradients
    bool vg = (W2 & 7) == 0;
    for (int l = 0; l < 4; l++) vg = vg && (l >= levels || !gp.p[l] || (reinterpret_cast<uintptr_t>(gp.p[l]) & 15) == 0);
    auto fold4 = [&](size_t rowbase, int w2, bool row_ok) {
        if (!row_ok || w2 >= W2) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (!vg || w2 + 3 >= W2)
            return make_float4(fold_grad<T>(gp, levels, rowbase, w2, W2), w2 + 1 < W2 ? fold_grad<T>(gp, levels, rowbase, w2 + 1, W2) : 0.f,
                               w2 + 2 < W2 ? fold_grad<T>(gp, levels, rowbase, w2 + 2, W2) : 0.f,
                               w2 + 3 < W2 ? fold_grad<T>(gp, levels, rowbase, w2 + 3, W2) : 0.f);
        // W2 % 8 == 0 and w2 % 4 == 0: every level's window is complete and its loads are aligned; same sums as fold_grad
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gp.p[0]) r = ld4(reinterpret_cast<const T *>(gp.p[0]) + rowbase * W2 + w2, 4, true);
        if (levels > 1 && gp.p[1]) {
            const T *q = reinterpret_cast<const T *>(gp.p[1]) + rowbase * (W2 >> 1) + (w2 >> 1);
            const float q0 = ldf(q), q1 = ldf(q + 1);
            r.x += 0.5f * q0; r.y += 0.5f * q0; r.z += 0.5f * q1; r.w += 0.5f * q1;
        }
        if (levels > 2 && gp.p[2]) {
            const float q0 = ldf(reinterpret_cast<const T *>(gp.p[2]) + rowbase * (W2 >> 2) + (w2 >> 2));
            r.x += 0.25f * q0; r.y += 0.25f * q0; r.z += 0.25f * q0; r.w += 0.25f * q0;
        }
        if (levels > 3 && gp.p[3]) {
            const float q0 = ldf(reinterpret_cast<const T *>(gp.p[3]) + rowbase * (W2 >> 3) + (w2 >> 3));
            r.x += 0.125f * q0; r.y += 0.125f * q0; r.z += 0.125f * q0; r.w += 0.125f * q0;
        }
        return r;
    };
    if (WRT_F1) {  // B(k = w2, n = w1): 4 along k = 4 consecutive w2 of row w1
        tile_gemm<BWD_MT, 1, true, true>(Wk, fa, [&](int k, int nn) { return fold4((size_t)g * W1 + n0 + nn, k, n0 + nn < W1); }, As, Bs, acc);
    } else {       // B(k = w1, n = w2): 4 along n = 4 consecutive w2 of row k
        tile_gemm<BWD_MT, 1, true, false>(Wk, fa, [&](int k, int nn) { return fold4((size_t)g * W1 + k, n0 + nn, k < W1); }, As, Bs, acc);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = n0 + (wid >> 1) * 32 + (lane & 31);
    T *o = gout + ((size_t)n * D * H + h) * Wn;
#pragma unroll
    for (int i = 0; i < BWD_MT; i++) {
        const int row0 = m0 + 64 * i + (wid & 1) * 32 + (lane >> 5) * 4;
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int d = row0 + (v >> 2) * 8 + (v & 3);
            if (d < D && col < Wn) stf(o + (size_t)d * H * Wn + col, acc[i][0][v] * scale);
        }
    }
}

// ---- fused multi-level lookup -------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <typename T>
__global__ __launch_bounds__(256) void k_lookup_fwd(PyrConstPtrs pyr, const float *__restrict__ coords, T *__restrict__ out, int total, int H1,
                                                    int W1, int W2, int levels, int r) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // (n*H1 + y)*W1 + x
    if (idx >= total) return;
    const int hw = H1 * W1;
    const int n = idx / hw, yx = idx - n * hw;
    const int rd = 2 * r + 1;
    const float x = coords[idx];
    int wl = W2;
    float inv = 1.f;
    for (int l = 0; l < levels; l++) {
        const float x0 = x * inv;  // coords / 2^l is exact
        const float fl = floorf(x0);
        const float dx = x0 - fl;
        const int xs = (int)fl - r;
        const T *v = reinterpret_cast<const T *>(pyr.p[l]) + (size_t)idx * wl;
        float prev = (xs >= 0 && xs < wl) ? ldf(v + xs) : 0.f;
        T *o = out + ((size_t)n * levels * rd + (size_t)l * rd) * hw + yx;
        for (int k = 0; k < rd; k++) {
            const int x1 = xs + k + 1;
            const float next = (x1 >= 0 && x1 < wl) ? ldf(v + x1) : 0.f;
            stf(o + (size_t)k * hw, prev * (1.0f - dx) + next * dx);
            prev = next;
        }
        wl >>= 1;
        inv *= 0.5f;
    }
}

// one thread per level-0 COLUMN position e = (row idx, c): writes element c of every level whose row is at least c+1 wide
// (each gradient row is owned by one (n, y, x), so there is no scatter and the zero fill is fused)
template <typename T>
__global__ __launch_bounds__(256) void k_lookup_bwd(const float *__restrict__ coords, const T *__restrict__ grad_out, PyrPtrs gpyr, size_t total,
                                                    int H1, int W1, int W2, int levels, int r) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % W2);
    const size_t idx = e / W2;
    const int hw = H1 * W1;
    const int n = (int)(idx / hw), yx = (int)(idx - (size_t)n * hw);
    const int rd = 2 * r + 1;
    const float x = coords[idx];
    int wl = W2;
    float inv = 1.f;
    for (int l = 0; l < levels; l++) {
        if (c < wl) {
            const float x0 = x * inv;
            const float fl = floorf(x0);
            const float dx = x0 - fl;
            const int i = c - ((int)fl - r);  // tap index of this element, valid 0..rd
            float gsum = 0.f;
            if (i >= 0 && i <= rd) {
                const T *go = grad_out + ((size_t)n * levels * rd + (size_t)l * rd) * hw + yx;
                if (i > 0) gsum += ldf(go + (size_t)(i - 1) * hw) * dx;
                if (i < rd) gsum += ldf(go + (size_t)i * hw) * (1.0f - dx);
            }
            stf(reinterpret_cast<T *>(gpyr.p[l]) + idx * wl + c, gsum);
        }
        wl >>= 1;
        inv *= 0.5f;
    }
}

// ---- convex upsampling --------------------------------------------------------------------------------------------------
// out[n, c, h f + i, w f + j] = sum_k softmax_k(mask[n, k f^2 + i f + j, h, w]) * f * flow[n, c, h + k / 3 - 1, w + k % 3 - 1]   (zero padded)
// One thread per (n, h, w, i, j) with the fine column fastest: mask reads walk j (stride H W: the 9 f^2 planes are read
// once, coalesced along w for fixed (i, j) by the neighbouring threads of other w), output stores are coalesced.
__global__ __launch_bounds__(256) void k_up_fwd(const float *__restrict__ flow, const float *__restrict__ mask, float *__restrict__ out, int N, int C,
                                                int H, int W, int f) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((n * H f + y) * W f + x)
    const int WF = W * f, HF = H * f;
    const size_t total = (size_t)N * HF * WF;
    if (t >= total) return;
    const int x = (int)(t % WF), y = (int)((t / WF) % HF), n = (int)(t / ((size_t)WF * HF));
    const int w = x / f, j = x - w * f, h = y / f, i = y - h * f;
    const size_t hw = (size_t)H * W;
    const float *m = mask + ((size_t)n * 9 * f * f + (size_t)i * f + j) * hw + (size_t)h * W + w;
    float lg[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        lg[k] = m[(size_t)k * f * f * hw];
        mx = fmaxf(mx, lg[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        lg[k] = expf(lg[k] - mx);
        den += lg[k];
    }
    const float inv = 1.f / den;
    for (int c = 0; c < C; c++) {
        const float *fl = flow + ((size_t)n * C + c) * hw;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
            const float v = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? (float)f * fl[(size_t)hh * W + ww] : 0.f;
            s += (lg[k] * inv) * v;
        }
        out[(((size_t)n * C + c) * HF + y) * WF + x] = s;
    }
}

// Backward.  Per fine pixel: p = softmax(9 logits), u_k = sum_c g_c * (f * flow_c at tap k); dlogit_k = p_k (u_k - sum_j p_j u_j).
// The flow gradient needs, per coarse cell and tap, S[k][c] = sum over the cell's f^2 fine pixels of p_k g_c; the cell (hh, ww)
// then gathers f * sum_k S[k][c] of its 9 neighbours (k_up_bwd_flow): fixed summation order, no atomics.
// Workgroup = 64 coarse columns x 4 groups that split the f^2 fine pixels: every mask / grad_mask access is coalesced along w.
__global__ __launch_bounds__(256) void k_up_bwd_cells(const float *__restrict__ flow, const float *__restrict__ mask, const float *__restrict__ gout,
                                                      float *__restrict__ gmask, float *__restrict__ S, int C, int H, int W, int f) {
    __shared__ float red[3][18][64];
    const int wq = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int w = blockIdx.x * 64 + wq, h = blockIdx.y, n = blockIdx.z;
    const bool in = w < W;
    const int WF = W * f, HF = H * f, ff = f * f;
    const size_t hw = (size_t)H * W;
    float tap[2][9], acc[2][9];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
            tap[c][k] = (in && c < C && hh >= 0 && hh < H && ww >= 0 && ww < W) ? (float)f * flow[((size_t)n * C + c) * hw + (size_t)hh * W + ww] : 0.f;
            acc[c][k] = 0.f;
        }
    if (in) {
        for (int ij = grp; ij < ff; ij += 4) {
            const int i = ij / f, j = ij - i * f;
            const size_t moff = ((size_t)n * 9 * ff + ij) * hw + (size_t)h * W + w;
            float p[9], mx = -INFINITY, den = 0.f;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                p[k] = mask[moff + (size_t)k * ff * hw];
                mx = fmaxf(mx, p[k]);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) {
                p[k] = expf(p[k] - mx);
                den += p[k];
            }
            const float inv = 1.f / den;
            const size_t goff = ((size_t)n * C * HF + (size_t)h * f + i) * WF + (size_t)w * f + j;
            const float g0 = gout[goff], g1 = C > 1 ? gout[goff + (size_t)HF * WF] : 0.f;
            float dot = 0.f, u[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                p[k] *= inv;
                u[k] = g0 * tap[0][k] + g1 * tap[1][k];
                dot += p[k] * u[k];
                acc[0][k] += p[k] * g0;
                acc[1][k] += p[k] * g1;
            }
            if (gmask) {
#pragma unroll
                for (int k = 0; k < 9; k++) gmask[moff + (size_t)k * ff * hw] = p[k] * (u[k] - dot);
            }
        }
    }
    if (!S) return;
    if (grp > 0) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int k = 0; k < 9; k++) red[grp - 1][c * 9 + k][wq] = acc[c][k];
    }
    __syncthreads();
    if (grp == 0 && in) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const float v = ((acc[c][k] + red[0][c * 9 + k][wq]) + red[1][c * 9 + k][wq]) + red[2][c * 9 + k][wq];
                if (c < C) S[(((size_t)n * 9 + k) * C + c) * hw + (size_t)h * W + w] = v;
            }
    }
}

__global__ __launch_bounds__(256) void k_up_bwd_flow(const float *__restrict__ S, float *__restrict__ gflow, int N, int C, int H, int W, int f) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((n * C + c) * H + hh) * W + ww
    const size_t hw = (size_t)H * W;
    if (t >= (size_t)N * C * hw) return;
    const int ww = (int)(t % W), hh = (int)((t / W) % H), c = (int)((t / hw) % C), n = (int)(t / (hw * C));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int h = hh - (k / 3 - 1), w = ww - (k % 3 - 1);
        if (h >= 0 && h < H && w >= 0 && w < W) s += S[(((size_t)n * 9 + k) * C + c) * hw + (size_t)h * W + w];
    }
    gflow[t] = (float)f * s;
}

}  // namespace

static bool dims_ok(int N, int D, int H, int W1, int W2, int levels, int dtype) {
    return N >= 0 && D >= 0 && H >= 0 && W1 >= 0 && W2 >= 0 && levels >= 1 && levels <= 4 && (dtype == 0 || dtype == 1);
}

extern "C" int cv_build_forward(const void *fmap1, const void *fmap2, void *const *pyramid, int N, int D, int H, int W1, int W2, int levels, int dtype,
                                void *stream) {
    if (!dims_ok(N, D, H, W1, W2, levels, dtype) || !pyramid) return GPSGS_E_INVALID;
    if ((size_t)N * H == 0 || W1 == 0 || W2 == 0) return GPSGS_OK;
    if (!fmap1 || !fmap2 || (size_t)N * H > 65535u) return GPSGS_E_INVALID;
    PyrPtrs pp;
    for (int l = 0; l < 4; l++) {
        pp.p[l] = l < levels ? pyramid[l] : nullptr;
        if (l < levels && !pp.p[l] && (W2 >> l) > 0) return GPSGS_E_INVALID;
    }
    const dim3 grid((W2 + 63) / 64, (W1 + 64 * FWD_MT - 1) / (64 * FWD_MT), N * H);
    const float scale = 1.0f / sqrtf((float)D);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(k_cv_fwd<float>, grid, dim3(256), 0, s, (const float *)fmap1, (const float *)fmap2, pp, D, H, W1, W2, levels, scale);
    else
        hipLaunchKernelGGL(k_cv_fwd<__half>, grid, dim3(256), 0, s, (const __half *)fmap1, (const __half *)fmap2, pp, D, H, W1, W2, levels, scale);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int cv_build_backward(const void *fmap1, const void *fmap2, const void *const *grad_pyramid, void *grad_fmap1, void *grad_fmap2, int N, int D,
                                 int H, int W1, int W2, int levels, int dtype, void *stream) {
    if (!dims_ok(N, D, H, W1, W2, levels, dtype) || !grad_pyramid) return GPSGS_E_INVALID;
    if ((size_t)N * H == 0 || D == 0) return GPSGS_OK;
    if (!fmap1 || !fmap2 || (!grad_fmap1 && !grad_fmap2) || (size_t)N * H > 65535u) return GPSGS_E_INVALID;
    PyrConstPtrs gp;
    for (int l = 0; l < 4; l++) gp.p[l] = l < levels ? grad_pyramid[l] : nullptr;
    const float scale = 1.0f / sqrtf((float)D);
    hipStream_t s = (hipStream_t)stream;
    if (grad_fmap1 && W1 > 0) {
        const dim3 grid((W1 + 63) / 64, (D + 64 * BWD_MT - 1) / (64 * BWD_MT), N * H);
        if (dtype == 0)
            hipLaunchKernelGGL((k_cv_bwd<float, true>), grid, dim3(256), 0, s, (const float *)fmap2, gp, (float *)grad_fmap1, D, H, W1, W2, levels, scale);
        else
            hipLaunchKernelGGL((k_cv_bwd<__half, true>), grid, dim3(256), 0, s, (const __half *)fmap2, gp, (__half *)grad_fmap1, D, H, W1, W2, levels, scale);
    }
    if (grad_fmap2 && W2 > 0) {
        const dim3 grid((W2 + 63) / 64, (D + 64 * BWD_MT - 1) / (64 * BWD_MT), N * H);
        if (dtype == 0)
            hipLaunchKernelGGL((k_cv_bwd<float, false>), grid, dim3(256), 0, s, (const float *)fmap1, gp, (float *)grad_fmap2, D, H, W1, W2, levels, scale);
        else
            hipLaunchKernelGGL((k_cv_bwd<__half, false>), grid, dim3(256), 0, s, (const __half *)fmap1, gp, (__half *)grad_fmap2, D, H, W1, W2, levels, scale);
    }
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int cs_lookup_forward(const void *const *pyramid, const float *coords, void *out, int N, int H1, int W1, int W2, int levels, int radius,
                                 int dtype, void *stream) {
    if (!dims_ok(N, 0, H1, W1, W2, levels, dtype) || radius < 0 || !pyramid) return GPSGS_E_INVALID;
    const long long total = (long long)N * H1 * W1;
    if (total == 0) return GPSGS_OK;
    if (!coords || !out || total > 0x7fffffffLL) return GPSGS_E_INVALID;
    PyrConstPtrs pp;
    for (int l = 0; l < 4; l++) {
        pp.p[l] = l < levels ? pyramid[l] : nullptr;
        if (l < levels && !pp.p[l] && (W2 >> l) > 0) return GPSGS_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_lookup_fwd<float>, grid, dim3(256), 0, s, pp, coords, (float *)out, (int)total, H1, W1, W2, levels, radius);
    else
        hipLaunchKernelGGL(k_lookup_fwd<__half>, grid, dim3(256), 0, s, pp, coords, (__half *)out, (int)total, H1, W1, W2, levels, radius);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int cs_lookup_backward(const float *coords, const void *grad_out, void *const *grad_pyramid, int N, int H1, int W1, int W2, int levels,
                                  int radius, int dtype, void *stream) {
    if (!dims_ok(N, 0, H1, W1, W2, levels, dtype) || radius < 0 || !grad_pyramid) return GPSGS_E_INVALID;
    const size_t total = (size_t)N * H1 * W1 * W2;
    if (total == 0) return GPSGS_OK;
    if (!coords || !grad_out) return GPSGS_E_INVALID;
    PyrPtrs pp;
    for (int l = 0; l < 4; l++) {
        pp.p[l] = l < levels ? grad_pyramid[l] : nullptr;
        if (l < levels && !pp.p[l] && (W2 >> l) > 0) return GPSGS_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_lookup_bwd<float>, grid, dim3(256), 0, s, coords, (const float *)grad_out, pp, total, H1, W1, W2, levels, radius);
    else
        hipLaunchKernelGGL(k_lookup_bwd<__half>, grid, dim3(256), 0, s, coords, (const __half *)grad_out, pp, total, H1, W1, W2, levels, radius);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int cu_upsample_forward(const float *flow, const float *mask, float *out, int N, int C, int H, int W, int factor, void *stream) {
    if (N < 0 || C < 1 || C > 4 || H < 0 || W < 0 || factor < 1 || factor > 16) return GPSGS_E_INVALID;
    const size_t total = (size_t)N * H * factor * W * factor;
    if (total == 0) return GPSGS_OK;
    if (!flow || !mask || !out) return GPSGS_E_INVALID;
    hipLaunchKernelGGL(k_up_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flow, mask, out, N, C, H, W, factor);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" size_t cu_upsample_scratch_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)N * 9 * C * H * W * sizeof(float);
}

extern "C" int cu_upsample_backward(const float *flow, const float *mask, const float *grad_out, float *grad_flow, float *grad_mask, void *scratch,
                                    int N, int C, int H, int W, int factor, void *stream) {
    if (N < 0 || C < 1 || C > 2 || H < 0 || W < 0 || factor < 1 || factor > 16) return GPSGS_E_INVALID;
    const size_t total = (size_t)N * H * W;
    if (total == 0) return GPSGS_OK;
    if (!flow || !mask || !grad_out || (!grad_flow && !grad_mask) || (grad_flow && !scratch) || H > 65535 || N > 65535) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float *S = grad_flow ? static_cast<float *>(scratch) : nullptr;
    hipLaunchKernelGGL(k_up_bwd_cells, dim3((W + 63) / 64, H, N), dim3(256), 0, s, flow, mask, grad_out, grad_mask, S, C, H, W, factor);
    if (grad_flow) {
        const size_t tf = total * C;
        hipLaunchKernelGGL(k_up_bwd_flow, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, s, S, grad_flow, N, C, H, W, factor);
    }
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
