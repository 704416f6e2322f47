// fused_loss.hip -- fused L1 + SSIM loss, forward and backward, for gfx950.
//
// The step immediately after the rasteriser in stage-2 training: train_stage2.py:70-72 computes
//     0.8 * l1_loss(pred, gt) + 0.2 * (1 - ssim(pred, gt))          on [B,3,2048,2048]
// with /root/reference/lib/loss.py:36-83: ssim() = five depthwise 11x11 convolutions (zero padding, Gaussian window
// sigma 1.5) + ~15 elementwise kernels, and autograd replays all of it backwards.  Here:
//   k_loss_fwd   one 32x32 output tile per workgroup: the window is applied separably (11 + 11 taps instead of 121) to the five
//                moments x1, x2, x1^2, x2^2, x1 x2.  Horizontal pass: every thread reads 16 consecutive inputs of one halo row
//                STRAIGHT FROM GLOBAL MEMORY (four dword-aligned 16-byte loads; the 42x42 halo is L2 / L1 resident, a staging
//                tile in LDS cost a pass, a barrier and 15 KB that capped the CU at 3 workgroups) and slides the window over 4
//                adjacent outputs (14 values feed 44 taps); only the row-filtered moments go through LDS.  Vertical pass: the SSIM
//                map value and the three partial-derivative maps the backward needs are formed in registers, |x1 - x2|
//                is added, and per-workgroup partial sums are written (deterministic two-level reduction, no atomics);
//   k_loss_reduce  sums the partials in double -> {mean L1, mean SSIM};
//   k_loss_bwd   same tiling, but the 42x42 halo of the three derivative maps IS staged in LDS (read from global directly, as the
//                forward does, the twelve unaligned 16-byte loads per item made it slower: 125 -> 159 us at [4,3,1024,1024]); then
//                dL/dpred = gL1 * sign(x1 - x2) / N + gSSIM * (F*M1 + 2 x1 F*M2 + x2 F*M3) / N.
// HBM traffic: forward reads 8 B and writes 12 B per element, backward reads 20 B and writes 4 B -- against roughly 30
// full-tensor passes in the eager version.
#include "gsr_common.h"

namespace {

constexpr int TS = 32, R = 5, HS = TS + 2 * R;  // tile side, window radius, halo side (42)
constexpr int TST = HS + 2;                     // row stride of a staged tile (backward): 44 floats, so 16-byte row reads stay aligned
constexpr int XG = TS / 4;                      // every thread produces 4 ADJACENT outputs per pass from a sliding window:
                                                // 14 LDS values feed 4 x 11 taps (3.4x fewer LDS reads than one output per thread)

struct Win {
    float g[11];
};

__device__ __forceinline__ float block_sum_256(float v, float *red /*[4]*/) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// 16 consecutive floats of a row staged in LDS (the sliding window uses the first 14)
__device__ __forceinline__ void load_row16_lds(const float *row, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 f = reinterpret_cast<const float4 *>(row)[q];
        v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
}
struct __attribute__((aligned(4))) F4u {  // 16 bytes at dword alignment (gfx950 global_load_dwordx4 needs no more)
    float x, y, z, w;
};
// 16 consecutive floats img[gy][gx0 .. gx0 + 15] with the zero padding of conv2d(padding = 5) (the sliding window uses the first 14).
// fast: the whole span lies inside the row (decided per workgroup) -> four 16-byte loads; otherwise element by element.
__device__ __forceinline__ void load_row16(const float *__restrict__ img, int H, int W, int gy, int gx0, bool fast, float (&v)[16]) {
    const bool row_ok = gy >= 0 && gy < H;
    if (fast) {
        const F4u *src = reinterpret_cast<const F4u *>(img + (size_t)(row_ok ? gy : 0) * W + gx0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const F4u f = src[q];
            v[4 * q] = row_ok ? f.x : 0.f; v[4 * q + 1] = row_ok ? f.y : 0.f; v[4 * q + 2] = row_ok ? f.z : 0.f; v[4 * q + 3] = row_ok ? f.w : 0.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int gx = gx0 + k;
            const bool in = row_ok && gx >= 0 && gx < W;
            v[k] = in ? img[(size_t)gy * W + gx] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void k_loss_fwd(const float *__restrict__ x1g, const float *__restrict__ x2g, int H, int W, Win win,
                                                  float *__restrict__ m1, float *__restrict__ m2, float *__restrict__ m3,
                                                  float2 *__restrict__ partial) {
    __shared__ float h[5][HS][TS + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *x1p = x1g + plane, *x2p = x2g + plane;
    const bool fast = x0 >= R && x0 + TS + R + 2 <= W;  // every 16-float span of this tile's halo rows lies inside the image row
    for (int i = tid; i < HS * XG; i += 256) {  // horizontal pass: 42 rows x 8 groups of 4 columns
        const int ly = i / XG, xg = i - ly * XG;
        float p[16], q[16];
        load_row16(x1p, H, W, y0 + ly - R, x0 + 4 * xg - R, fast, p);
        load_row16(x2p, H, W, y0 + ly - R, x0 + 4 * xg - R, fast, q);
        float pp[14], qq[14], pq[14];
#pragma unroll
        for (int k = 0; k < 14; k++) { pp[k] = p[k] * p[k]; qq[k] = q[k] * q[k]; pq[k] = p[k] * q[k]; }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {  // same tap order and products as one-output-per-thread: bit-identical sums
                const float w = win.g[k];
                a += w * p[o + k]; b += w * q[o + k]; aa += w * pp[o + k]; bb += w * qq[o + k]; ab += w * pq[o + k];
            }
            const int lx = 4 * xg + o;
            h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = aa; h[3][ly][lx] = bb; h[4][ly][lx] = ab;
        }
    }
    __syncthreads();
    const int tx = tid & (TS - 1), yg = tid / TS;  // vertical pass: column tx, rows 4 yg .. 4 yg + 3
    float col[5][14];
#pragma unroll
    for (int k = 0; k < 14; k++)
#pragma unroll
        for (int c = 0; c < 5; c++) col[c][k] = h[c][4 * yg + k][tx];
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1v = 0.f, sv = 0.f;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.g[k];
            mu1 += w * col[0][o + k]; mu2 += w * col[1][o + k];
            e11 += w * col[2][o + k]; e22 += w * col[3][o + k]; e12 += w * col[4][o + k];
        }
        const int ty = 4 * yg + o, gx = x0 + tx, gy = y0 + ty;
        const bool in = gx < W && gy < H;
        const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
        const float rB1 = __builtin_amdgcn_rcpf(B1), rB2 = __builtin_amdgcn_rcpf(B2);  // 1-ulp reciprocals: B1, B2 >= C1, C2 > 0
        const float smap = A1 * A2 * rB1 * rB2;
        if (in && m1) {
            const float ds_dmu1 = 2.f * mu2 * A2 * rB1 * rB2 - 2.f * mu1 * smap * rB1;
            const float ds_ds1 = -smap * rB2;
            const float ds_ds12 = 2.f * A1 * rB1 * rB2;
            const size_t qi = plane + (size_t)gy * W + gx;
            m1[qi] = ds_dmu1 - 2.f * mu1 * ds_ds1 - mu2 * ds_ds12;  // total derivative w.r.t. the filtered mean
            m2[qi] = ds_ds1;
            m3[qi] = ds_ds12;
        }
        l1v += in ? fabsf(x1p[(size_t)gy * W + gx] - x2p[(size_t)gy * W + gx]) : 0.f;
        sv += in ? smap : 0.f;
    }
    const float sl = block_sum_256(l1v, red);
    const float ss = block_sum_256(sv, red);
    if (tid == 0) partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(sl, ss);
}

__global__ __launch_bounds__(1024) void k_loss_reduce(const float2 *__restrict__ partial, int n, double inv_count, float *__restrict__ out2) {
    __shared__ double r1[16], r2[16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {  // fixed order per thread, fixed tree below: deterministic
        const float2 p = partial[i];
        a += (double)p.x;
        b += (double)p.y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < 16; w++) { s1 += r1[w]; s2 += r2[w]; }
        out2[0] = (float)(s1 * inv_count);
        out2[1] = (float)(s2 * inv_count);
    }
}

__global__ __launch_bounds__(256) void k_loss_bwd(const float *__restrict__ x1g, const float *__restrict__ x2g, const float *__restrict__ m1,
                                                  const float *__restrict__ m2, const float *__restrict__ m3, int H, int W, Win win,
                                                  const float *__restrict__ g_out2 /* d/dL1mean, d/dSSIMmean (device) */, float inv_count,
                                                  float *__restrict__ dx1) {
    __shared__ __attribute__((aligned(16))) float t[3][HS][TST];
    __shared__ float h[3][HS][TS + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < HS * TST; i += 256) {
        const int ly = i / TST, lx = i - ly * TST, gy = y0 + ly - R, gx = x0 + lx - R;
        const bool in = lx < HS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t q = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
        t[0][ly][lx] = in ? m1[q] : 0.f;
        t[1][ly][lx] = in ? m2[q] : 0.f;
        t[2][ly][lx] = in ? m3[q] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < HS * XG; i += 256) {
        const int ly = i / XG, xg = i - ly * XG;
        float v[3][16];
#pragma unroll
        for (int c = 0; c < 3; c++) load_row16_lds(&t[c][ly][4 * xg], v[c]);
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = win.g[k];
                a += w * v[0][o + k]; b += w * v[1][o + k]; c += w * v[2][o + k];
            }
            const int lx = 4 * xg + o;
            h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = c;
        }
    }
    __syncthreads();
    const int tx = tid & (TS - 1), yg = tid / TS;
    float col[3][14];
#pragma unroll
    for (int k = 0; k < 14; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) col[c][k] = h[c][4 * yg + k][tx];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        float f1 = 0.f, f2 = 0.f, f3 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.g[k];
            f1 += w * col[0][o + k]; f2 += w * col[1][o + k]; f3 += w * col[2][o + k];
        }
        const int gx = x0 + tx, gy = y0 + 4 * yg + o;
        if (gx < W && gy < H) {
            const size_t q = plane + (size_t)gy * W + gx;
            const float a = x1g[q], b = x2g[q];
            const float d = a - b;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch.abs backward: sign(0) = 0
            dx1[q] = (g_out2[0] * sgn + g_out2[1] * (f1 + 2.f * a * f2 + b * f3)) * inv_count;
        }
    }
}

Win make_window() {  // lib/loss.py:40-42: exp(-(x-5)^2 / (2 * 1.5^2)), normalised in fp32
    Win w;
    float g[11], s = 0.f;
    for (int x = 0; x < 11; x++) {
        g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5));
        s += g[x];
    }
    for (int x = 0; x < 11; x++) w.g[x] = g[x] / s;
    return w;
}

}  // namespace

extern "C" size_t fl_scratch_bytes(int planes, int H, int W) {
    if (planes < 0 || H < 0 || W < 0) return 0;
    return (size_t)planes * ((H + TS - 1) / TS) * ((W + TS - 1) / TS) * sizeof(float2) + 16;
}

extern "C" int fl_l1_ssim_forward(const float *pred, const float *gt, int planes, int H, int W, float *m1, float *m2, float *m3,
                                  void *scratch, float *out2, void *stream) {
    if (planes < 0 || H < 0 || W < 0) return GPSGS_E_INVALID;
    if ((size_t)planes * H * W == 0) return GPSGS_E_INVALID;  // a mean over nothing is undefined (torch returns nan)
    if (!pred || !gt || !scratch || !out2) return GPSGS_E_INVALID;
    if ((m1 || m2 || m3) && !(m1 && m2 && m3)) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    hipLaunchKernelGGL(k_loss_fwd, grid, dim3(256), 0, s, pred, gt, H, W, make_window(), m1, m2, m3, (float2 *)scratch);
    const int n = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, s, (const float2 *)scratch, n, 1.0 / ((double)planes * H * W), out2);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int fl_l1_ssim_backward(const float *pred, const float *gt, const float *m1, const float *m2, const float *m3, int planes, int H,
                                   int W, const float *grad_out2, float *d_pred, void *stream) {
    if (planes < 0 || H < 0 || W < 0) return GPSGS_E_INVALID;
    if ((size_t)planes * H * W == 0) return GPSGS_OK;
    if (!pred || !gt || !m1 || !m2 || !m3 || !grad_out2 || !d_pred) return GPSGS_E_INVALID;
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    hipLaunchKernelGGL(k_loss_bwd, grid, dim3(256), 0, (hipStream_t)stream, pred, gt, m1, m2, m3, H, W, make_window(), grad_out2,
                       (float)(1.0 / ((double)planes * H * W)), d_pred);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
