// pack_views.hip -- fused mask-compaction + AoS pack of the per-pixel Gaussian maps.
//
// Replaces, for a whole batch in three small launches and ZERO host syncs, what /root/reference/lib/GaussianRender.py:15-34
// does per sample with 10 boolean-mask gathers (each a device sync), 5 concats and an affine:
//     for view in (lmain, rmain):  x_i = map[i].permute(1,2,0).view(-1,C)[pts_valid[i]]
//     concat L then R;  rgb = rgb * 0.5 + 0.5
// Row order is exactly the reference's: sample b, then view, then valid pixels in raster order.  The packed rows of all
// samples live back to back; sample_offsets[b] .. sample_offsets[b+1] delimits sample b (kept on the device; the host
// reads B+1 integers once if it needs per-sample tensors).
//
//   k_pack_count   valid pixels per 1024-pixel block                       (reads 1 B/pixel)
//   k_pack_scan    one workgroup: exclusive scan of the block counts + per-sample offsets
//   k_pack_gather  row = block offset + rank inside the block (wave ballots); reads the 5 maps with arbitrary element
//                  strides (the reference's xyz is a permuted view), writes AoS rows + row_of_pixel for the backward
//   k_pack_bwd     per pixel: planar gradients of the maps from the packed row gradients (zero where invalid): the
//                  fill and the scatter are one pass, no memset, no atomics
// HBM-bound: 57 B read per pixel + 56 B written per valid pixel.
#include "gsr_common.h"

#pragma clang fp contract(off)  // rgb = img * 0.5 + 0.5 must round like torch's separate mul and add

namespace {

constexpr int PB = 1024;  // pixels per block
constexpr int MAXV = 4;

struct PackArgs {
    int B, V, S2, nblk;
    GsrStrided valid[MAXV], xyz[MAXV], img[MAXV], rot[MAXV], scale[MAXV], opacity[MAXV];
};

__device__ __forceinline__ bool is_valid(const PackArgs &a, int b, int v, int pix) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(a.valid[v].ptr);
    return pix < a.S2 && p[(int64_t)b * a.valid[v].batch_stride + (int64_t)pix * a.valid[v].pixel_stride] != 0;
}

__global__ __launch_bounds__(PB) void k_pack_count(PackArgs a, uint32_t *__restrict__ counts) {
    __shared__ uint32_t red[PB / 64];
    const int blk = blockIdx.x, v = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const bool ok = is_valid(a, b, v, blk * PB + tid);
    const unsigned long long m = __ballot(ok);
    if ((tid & 63) == 0) red[tid >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (int w = 0; w < PB / 64; w++) t += red[w];
        counts[((size_t)b * a.V + v) * a.nblk + blk] = t;
    }
}

// exclusive scan over the flattened (b, v, blk) order, in place; offsets[b] = first row of sample b, offsets[B] = total
__global__ __launch_bounds__(1024) void k_pack_scan(uint32_t *__restrict__ counts, int n, int per_sample, int B,
                                                    uint32_t *__restrict__ sample_offsets) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n ? counts[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const uint32_t ex = carry_s + woff + x - v;
        if (i < n) {
            counts[i] = ex;
            if (i % per_sample == 0) sample_offsets[i / per_sample] = ex;
        }
        __syncthreads();
        if (tid == 1023) carry_s = ex + v;
        __syncthreads();
    }
    if (tid == 0) sample_offsets[B] = carry_s;
}

__device__ __forceinline__ float ldm(const GsrStrided &s, int b, int pix, int c) {
    return reinterpret_cast<const float *>(s.ptr)[(int64_t)b * s.batch_stride + (int64_t)pix * s.pixel_stride + (int64_t)c * s.channel_stride];
}

__global__ __launch_bounds__(PB) void k_pack_gather(PackArgs a, const uint32_t *__restrict__ block_offset, float *__restrict__ o_xyz,
                                                    float *__restrict__ o_rgb, float *__restrict__ o_rot, float *__restrict__ o_scale,
                                                    float *__restrict__ o_op, uint32_t *__restrict__ row_of_pixel) {
    __shared__ uint32_t wcnt[PB / 64];
    const int blk = blockIdx.x, v = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pix = blk * PB + tid;
    const bool ok = is_valid(a, b, v, pix);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wcnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wcnt[w];
    const uint32_t row = block_offset[((size_t)b * a.V + v) * a.nblk + blk] + woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pix < a.S2) row_of_pixel[((size_t)b * a.V + v) * a.S2 + pix] = ok ? row : 0xffffffffu;
    if (!ok) return;
    const size_t r3 = (size_t)row * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o_xyz[r3 + c] = ldm(a.xyz[v], b, pix, c);
        o_rgb[r3 + c] = ldm(a.img[v], b, pix, c) * 0.5f + 0.5f;
        o_scale[r3 + c] = ldm(a.scale[v], b, pix, c);
    }
    *reinterpret_cast<float4 *>(o_rot + (size_t)row * 4) =
        make_float4(ldm(a.rot[v], b, pix, 0), ldm(a.rot[v], b, pix, 1), ldm(a.rot[v], b, pix, 2), ldm(a.rot[v], b, pix, 3));
    o_op[row] = ldm(a.opacity[v], b, pix, 0);
}

struct PackBwdArgs {
    int B, V, S2;
    float *d_xyz[MAXV], *d_img[MAXV], *d_rot[MAXV], *d_scale[MAXV], *d_op[MAXV];  // contiguous [B,S2,3] / [B,C,S2]; any may be NULL
};

__global__ __launch_bounds__(256) void k_pack_bwd(PackBwdArgs a, const uint32_t *__restrict__ row_of_pixel, const float *__restrict__ g_xyz,
                                                  const float *__restrict__ g_rgb, const float *__restrict__ g_rot,
                                                  const float *__restrict__ g_scale, const float *__restrict__ g_op) {
    const int pix = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, b = blockIdx.z;
    if (pix >= a.S2) return;
    const uint32_t row = row_of_pixel[((size_t)b * a.V + v) * a.S2 + pix];
    const bool ok = row != 0xffffffffu;
    const size_t S2 = (size_t)a.S2;
    if (a.d_xyz[v]) {  // [B, S2, 3]
        float *d = a.d_xyz[v] + ((size_t)b * S2 + pix) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) d[c] = (ok && g_xyz) ? g_xyz[(size_t)row * 3 + c] : 0.f;
    }
    if (a.d_img[v])
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_img[v][((size_t)b * 3 + c) * S2 + pix] = (ok && g_rgb) ? g_rgb[(size_t)row * 3 + c] * 0.5f : 0.f;
    if (a.d_scale[v])
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_scale[v][((size_t)b * 3 + c) * S2 + pix] = (ok && g_scale) ? g_scale[(size_t)row * 3 + c] : 0.f;
    if (a.d_rot[v])
#pragma unroll
        for (int c = 0; c < 4; c++) a.d_rot[v][((size_t)b * 4 + c) * S2 + pix] = (ok && g_rot) ? g_rot[(size_t)row * 4 + c] : 0.f;
    if (a.d_op[v]) a.d_op[v][(size_t)b * S2 + pix] = (ok && g_op) ? g_op[row] : 0.f;
}

}  // namespace

extern "C" size_t gsr_pack_scratch_bytes(int B, int n_views, int S2) {
    if (B < 0 || n_views < 0 || S2 < 0) return 0;
    const size_t nblk = ((size_t)S2 + PB - 1) / PB;
    return ((size_t)B * n_views * nblk + 1) * 4;
}

extern "C" int gsr_pack_views(int B, int n_views, int S2, const GsrStrided *valid, const GsrStrided *xyz, const GsrStrided *img,
                              const GsrStrided *rot, const GsrStrided *scale, const GsrStrided *opacity, float *out_xyz, float *out_rgb,
                              float *out_rot, float *out_scale, float *out_opacity, uint32_t *row_of_pixel, uint32_t *sample_offsets,
                              uint32_t *scratch, void *stream) {
    if (B < 0 || n_views < 1 || n_views > MAXV || S2 < 0) return GPSGS_E_INVALID;
    if (!sample_offsets) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0 || S2 == 0) return hipMemsetAsync(sample_offsets, 0, (size_t)(B + 1) * 4, s) == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
    if (!valid || !xyz || !img || !rot || !scale || !opacity || !out_xyz || !out_rgb || !out_rot || !out_scale || !out_opacity ||
        !row_of_pixel || !scratch)
        return GPSGS_E_INVALID;
    PackArgs a;
    a.B = B; a.V = n_views; a.S2 = S2; a.nblk = (S2 + PB - 1) / PB;
    for (int v = 0; v < n_views; v++) {
        a.valid[v] = valid[v]; a.xyz[v] = xyz[v]; a.img[v] = img[v]; a.rot[v] = rot[v]; a.scale[v] = scale[v]; a.opacity[v] = opacity[v];
        if (!valid[v].ptr || !xyz[v].ptr || !img[v].ptr || !rot[v].ptr || !scale[v].ptr || !opacity[v].ptr) return GPSGS_E_INVALID;
    }
    const dim3 grid(a.nblk, n_views, B);
    hipLaunchKernelGGL(k_pack_count, grid, dim3(PB), 0, s, a, scratch);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, scratch, B * n_views * a.nblk, n_views * a.nblk, B, sample_offsets);
    hipLaunchKernelGGL(k_pack_gather, grid, dim3(PB), 0, s, a, scratch, out_xyz, out_rgb, out_rot, out_scale, out_opacity, row_of_pixel);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int gsr_pack_views_backward(int B, int n_views, int S2, const uint32_t *row_of_pixel, const float *g_xyz, const float *g_rgb,
                                       const float *g_rot, const float *g_scale, const float *g_opacity, float *const *d_xyz,
                                       float *const *d_img, float *const *d_rot, float *const *d_scale, float *const *d_opacity,
                                       void *stream) {
    if (B < 0 || n_views < 1 || n_views > MAXV || S2 < 0) return GPSGS_E_INVALID;
    if (B == 0 || S2 == 0) return GPSGS_OK;
    if (!row_of_pixel) return GPSGS_E_INVALID;
    PackBwdArgs a;
    a.B = B; a.V = n_views; a.S2 = S2;
    for (int v = 0; v < MAXV; v++) {
        const bool in = v < n_views;
        a.d_xyz[v] = (in && d_xyz) ? d_xyz[v] : nullptr;
        a.d_img[v] = (in && d_img) ? d_img[v] : nullptr;
        a.d_rot[v] = (in && d_rot) ? d_rot[v] : nullptr;
        a.d_scale[v] = (in && d_scale) ? d_scale[v] : nullptr;
        a.d_op[v] = (in && d_opacity) ? d_opacity[v] : nullptr;
    }
    hipLaunchKernelGGL(k_pack_bwd, dim3((S2 + 255) / 256, n_views, B), dim3(256), 0, (hipStream_t)stream, a, row_of_pixel, g_xyz, g_rgb, g_rot,
                       g_scale, g_opacity);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
