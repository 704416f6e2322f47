// gsr_composite.hip -- forward and backward alpha compositing for gfx950 (one 256-thread workgroup per 16x16 tile).
//
// Semantics: SURVEY.md section 9.2 (front-to-back blend with the power>0 / alpha<1/255 / T<1e-4 rules) and section 9.3 (back-to-
// front gradient recurrence); replaces upstream renderCUDA forward/backward (called through
// /root/reference/gaussian_renderer/__init__.py:54-62 and its autograd backward).
//
// CDNA4 mapping:
//   * 4 wave64 per tile; each wave owns a compact 8x8 pixel block (lane = 8x8 raster), so a splat of a few pixels
//     radius is skipped by whole waves (s_cbranch_execz) instead of by scattered lanes;
//   * every round stages 256 splat records {x,y,A,B | C,op,r,g | b} -- colour included -- into LDS with one 48-byte
//     gather per thread; the inner loop then reads wave-uniform LDS addresses (broadcast, conflict free);
//   * backward: per-(pixel, splat) gradient terms are summed across the wave with DPP row/bank shifts
//     (v_add_f32 dpp, no LDS traffic), accumulated per staged splat in LDS by one lane per wave, and flushed with
//     ONE global atomic per gradient component per (tile, splat) -- upstream issues one per (pixel, splat).
#include "gsr_common.h"

namespace {

constexpr int BATCH = 256;

__device__ __forceinline__ void tile_pixel(int tid, int &lx, int &ly) {
    const int w = tid >> 6, lane = tid & 63;
    lx = ((w & 1) << 3) | (lane & 7);
    ly = ((w >> 1) << 3) | (lane >> 3);
}

__global__ __launch_bounds__(256) void k_composite_fwd(int W, int H, int gx, const GsrSplat *__restrict__ splats,
                                                       const uint32_t *__restrict__ tile_offset,
                                                       const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                       float *__restrict__ out_color, float *__restrict__ final_T,
                                                       uint32_t *__restrict__ n_contrib, const GsrHeader *__restrict__ hdr) {
    __shared__ float4 sA[BATCH];
    __shared__ float4 sB[BATCH];
    __shared__ float sC[BATCH];
    if (hdr->overflow) return;
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    int lx, ly;
    tile_pixel(tid, lx, ly);
    const int px = tx * GSR_TILE + lx, py = ty * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = tile_offset[tile], r1 = tile_offset[tile + 1];

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;
    for (uint32_t base = r0; base < r1; base += BATCH) {
        if (__syncthreads_count(done) == BATCH) break;
        const uint32_t k = base + tid;
        if (k < r1) {
            const float4 *s = reinterpret_cast<const float4 *>(splats + point_list[k]);
            sA[tid] = s[0];
            sB[tid] = s[1];
            sC[tid] = s[2].x;
        }
        __syncthreads();
        const int cnt = (int)min((uint32_t)BATCH, r1 - base);
        for (int j = 0; !done && j < cnt; j++) {
            contributor++;
            const float4 a = sA[j];
            const float4 b = sB[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.f) continue;
            const float alpha = fminf(0.99f, b.y * __expf(power));
            if (alpha < 1.f / 255.f) continue;
            const float test_T = T * (1.f - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            const float w = alpha * T;
            C0 += b.z * w;
            C1 += b.w * w;
            C2 += sC[j] * w;
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        const size_t npix = (size_t)W * H, q = (size_t)py * W + px;
        final_T[q] = T;
        n_contrib[q] = last;
        out_color[q] = C0 + T * bg[0];
        out_color[npix + q] = C1 + T * bg[1];
        out_color[2 * npix + q] = C2 + T * bg[2];
    }
}

// ---- wave64 sum via DPP; result valid in lane 63 -----------------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xF, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF, 0xF>(v);  // row_shr:8   -> lanes 12..15 of each row hold the row sum
    v = dpp_add<0x142, 0xA, 0xF>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC, 0xF>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = wave sum
    return v;
}

__global__ __launch_bounds__(256) void k_composite_bwd(int W, int H, int gx, const GsrSplat *__restrict__ splats,
                                                       const uint32_t *__restrict__ tile_offset,
                                                       const uint32_t *__restrict__ point_list, const float *__restrict__ bg,
                                                       const float *__restrict__ dL_dpix, const float *__restrict__ final_T,
                                                       const uint32_t *__restrict__ n_contrib, GsrGradAcc *__restrict__ gacc,
                                                       const GsrHeader *__restrict__ hdr) {
    __shared__ float4 sA[BATCH];
    __shared__ float4 sB[BATCH];
    __shared__ float sC[BATCH];
    __shared__ uint32_t sId[BATCH];
    __shared__ float sAcc[BATCH * 9];
    __shared__ uint32_t sMax[4];
    if (hdr->overflow) return;
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = tile_offset[tile], r1 = tile_offset[tile + 1];
    if (r1 <= r0) return;
    int lx, ly;
    tile_pixel(tid, lx, ly);
    const int px = tx * GSR_TILE + lx, py = ty * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t npix = (size_t)W * H, q = (size_t)py * W + px;

    const float T_final = inside ? final_T[q] : 0.f;
    const uint32_t last = inside ? n_contrib[q] : 0u;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (inside) {
        d0 = dL_dpix[q];
        d1 = dL_dpix[npix + q];
        d2 = dL_dpix[2 * npix + q];
    }
    const float bg_dot = bg[0] * d0 + bg[1] * d1 + bg[2] * d2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // deepest contributor over the tile: nothing behind it receives gradient
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) sMax[wid] = m;
    __syncthreads();
    const uint32_t max_last = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));

    float T = T_final, ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

    // positions are 0-based from the front of the tile list; walk from max_last-1 down to 0 in rounds of 256
    for (int64_t top = (int64_t)max_last - 1; top >= 0; top -= BATCH) {
        __syncthreads();  // previous round's flush has finished with sAcc / sId
        const int cnt = (int)min((int64_t)BATCH, top + 1);
        if (tid < cnt) {
            const uint32_t id = point_list[r0 + (uint32_t)(top - tid)];
            const float4 *s = reinterpret_cast<const float4 *>(splats + id);
            sA[tid] = s[0];
            sB[tid] = s[1];
            sC[tid] = s[2].x;
            sId[tid] = id;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) sAcc[tid * 9 + k] = 0.f;
        __syncthreads();
        for (int j = 0; j < cnt; j++) {
            const uint32_t pos = (uint32_t)(top - j);
            const float4 a = sA[j];
            const float4 b = sB[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, b.y * G);
            const bool valid = (pos < last) && !(power > 0.f) && !(alpha < 1.f / 255.f);
            if (!__any(valid)) continue;  // wave-uniform

            float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_mx = 0.f, g_my = 0.f, g_xx = 0.f, g_xy = 0.f, g_yy = 0.f, g_op = 0.f;
            if (valid) {
                const float rcp = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * rcp;
                const float dchannel_dcolor = alpha * T;
                const float c0 = b.z, c1 = b.w, c2 = sC[j];
                ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0;
                ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1;
                ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2;
                lc0 = c0; lc1 = c1; lc2 = c2;
                float dL_dalpha = (c0 - ar0) * d0 + (c1 - ar1) * d1 + (c2 - ar2) * d2;
                g_r = dchannel_dcolor * d0;
                g_g = dchannel_dcolor * d1;
                g_b = dchannel_dcolor * d2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * rcp) * bg_dot;
                const float dL_dG = b.y * dL_dalpha;  // straight through the 0.99 clamp, like upstream
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_xx = -0.5f * gdx * dx * dL_dG;
                g_xy = -0.5f * gdx * dy * dL_dG;
                g_yy = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            g_r = wave_sum_to_lane63(g_r);
            g_g = wave_sum_to_lane63(g_g);
            g_b = wave_sum_to_lane63(g_b);
            g_mx = wave_sum_to_lane63(g_mx);
            g_my = wave_sum_to_lane63(g_my);
            g_xx = wave_sum_to_lane63(g_xx);
            g_xy = wave_sum_to_lane63(g_xy);
            g_yy = wave_sum_to_lane63(g_yy);
            g_op = wave_sum_to_lane63(g_op);
            if (lane == 63) {
                float *acc = sAcc + j * 9;
                atomicAdd(acc + 0, g_r); atomicAdd(acc + 1, g_g); atomicAdd(acc + 2, g_b);
                atomicAdd(acc + 3, g_mx); atomicAdd(acc + 4, g_my);
                atomicAdd(acc + 5, g_xx); atomicAdd(acc + 6, g_xy); atomicAdd(acc + 7, g_yy);
                atomicAdd(acc + 8, g_op);
            }
        }
        __syncthreads();
        if (tid < cnt) {
            const float *acc = sAcc + tid * 9;
            float *dst = reinterpret_cast<float *>(gacc + sId[tid]);
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const float v = acc[k];
                if (v != 0.f) atomicAdd(dst + k, v);
            }
        }
    }
}

}  // namespace

void gsr_launch_composite_fwd(int W, int H, int gx, int gy, const GsrSplat *splats, const uint32_t *tile_offset, const uint32_t *point_list,
                              const float *bg, float *out_color, float *final_T, uint32_t *n_contrib, const GsrHeader *hdr, hipStream_t s) {
    if (gx * gy <= 0) return;
    hipLaunchKernelGGL(k_composite_fwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, splats, tile_offset, point_list, bg, out_color, final_T,
                       n_contrib, hdr);
}

void gsr_launch_composite_bwd(int W, int H, int gx, int gy, const GsrSplat *splats, const uint32_t *tile_offset, const uint32_t *point_list,
                              const float *bg, const float *dL_dpix, const float *final_T, const uint32_t *n_contrib, GsrGradAcc *gacc,
                              const GsrHeader *hdr, hipStream_t s) {
    if (gx * gy <= 0) return;
    hipLaunchKernelGGL(k_composite_bwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, splats, tile_offset, point_list, bg, dL_dpix, final_T,
                       n_contrib, gacc, hdr);
}
