// unproject.hip -- fused disparity -> inverse depth -> world points, forward and backward.
//
// Replaces /root/reference/lib/utils.py:113-120 (flow2depth), :88-110 (depth2pc) and the validity test of
// /root/reference/lib/network.py:66-69 (about a dozen elementwise kernels + two bmm per view) with one pass per direction:
//     depth = -(ref_cx - cx - flow) / Tf_x * mask                    (this "depth" is INVERSE depth)
//     z = 1 / (depth + 1e-8);  X = (u + 0.5 - cx) z / fx;  Y = (v + 0.5 - cy) z / fy;  xyz = R^T [X,Y,z] - R^T t
//     valid = depth != 0
// Backward: d_flow = (g_depth - z^2 * g_xyz . R^T[ax, ay, 1]) * mask / Tf_x with ax = (u + 0.5 - cx) / fx.
// HBM-bound: 8 B read + 17 B written per pixel forward.  Evaluated without FMA contraction in the reference's operation
// order: depth and valid are bit-identical to torch's, xyz differs only by the summation order of the 3x3 product.
#include "gsr_common.h"

#pragma clang fp contract(off)

namespace {

struct UnprojCam {  // per batch element, host-filled
    float offset, tf, fx, fy, cx, cy;
    float Rt[9];   // R^T row-major
    float Rtt[3];  // R^T t
};
constexpr int MAXB = 16;
struct UnprojArgs {
    int B, S;
    UnprojCam cam[MAXB];
};

__device__ __forceinline__ UnprojCam cam_from_device(const float *__restrict__ cams, int b);

__global__ __launch_bounds__(256) void k_unproject_fwd(UnprojArgs a, const float *__restrict__ cams_dev, const float *__restrict__ flow, const float *__restrict__ mask,
                                                      int64_t mask_bstride, float *__restrict__ depth, float *__restrict__ xyz,
                                                      uint8_t *__restrict__ valid) {
    const int S2 = a.S * a.S, pix = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (pix >= S2) return;
    const UnprojCam c = cams_dev ? cam_from_device(cams_dev, b) : a.cam[b];
    const int v = pix / a.S, u = pix - v * a.S;
    const float disparity = c.offset - flow[(size_t)b * S2 + pix];
    float d = -disparity / c.tf;
    d = d * mask[(size_t)b * mask_bstride + pix];
    depth[(size_t)b * S2 + pix] = d;
    valid[(size_t)b * S2 + pix] = d != 0.0f;
    const float z = 1.0f / (d + 1e-8f);
    const float X = ((float)u + 0.5f - c.cx) * z / c.fx;
    const float Y = ((float)v + 0.5f - c.cy) * z / c.fy;
    float *o = xyz + ((size_t)b * S2 + pix) * 3;
    o[0] = (c.Rt[0] * X + c.Rt[1] * Y + c.Rt[2] * z) - c.Rtt[0];
    o[1] = (c.Rt[3] * X + c.Rt[4] * Y + c.Rt[5] * z) - c.Rtt[1];
    o[2] = (c.Rt[6] * X + c.Rt[7] * Y + c.Rt[8] * z) - c.Rtt[2];
}

__global__ __launch_bounds__(256) void k_unproject_bwd(UnprojArgs a, const float *__restrict__ cams_dev, const float *__restrict__ depth, const float *__restrict__ mask,
                                                      int64_t mask_bstride, const float *__restrict__ g_depth,
                                                      const float *__restrict__ g_xyz, int64_t gx_b, int64_t gx_p, int64_t gx_c,
                                                      float *__restrict__ d_flow) {
    const int S2 = a.S * a.S, pix = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (pix >= S2) return;
    const UnprojCam c = cams_dev ? cam_from_device(cams_dev, b) : a.cam[b];
    const int v = pix / a.S, u = pix - v * a.S;
    const float d = depth[(size_t)b * S2 + pix];
    const float z = 1.0f / (d + 1e-8f);
    float g = g_depth ? g_depth[(size_t)b * S2 + pix] : 0.f;
    if (g_xyz) {
        const float ax = ((float)u + 0.5f - c.cx) / c.fx, ay = ((float)v + 0.5f - c.cy) / c.fy;
        const float *gp = g_xyz + (int64_t)b * gx_b + (int64_t)pix * gx_p;
        const float g0 = gp[0], g1 = gp[gx_c], g2 = gp[2 * gx_c];
        const float dir0 = c.Rt[0] * ax + c.Rt[1] * ay + c.Rt[2], dir1 = c.Rt[3] * ax + c.Rt[4] * ay + c.Rt[5],
                    dir2 = c.Rt[6] * ax + c.Rt[7] * ay + c.Rt[8];
        g += -(z * z) * (g0 * dir0 + g1 * dir1 + g2 * dir2);
    }
    d_flow[(size_t)b * S2 + pix] = g * mask[(size_t)b * mask_bstride + pix] / c.tf;
}

// The same per-sample constants from camera arrays that live in DEVICE memory: cams[b] = {ref_intr 3x3, intr 3x3, extr 3x4 row-major, Tf_x} = 31 floats
// (wave-uniform scalar loads; the arithmetic of fill() below, operation for operation, so both forms give the same bits).  A caller whose cameras
// already sit on the GPU (the reference moves every item of its data dict there, train_stage2.py:154-156) then never has to read them back: reading
// them on the host costs a device synchronisation in the MIDDLE of the network forward (measured in BASELINE config 3 with the import hook: 13.4
// instead of 17.2 views/s inside a sample, the host could no longer run ahead of the GPU).
constexpr int CAM_FLOATS = 31;
__device__ __forceinline__ UnprojCam cam_from_device(const float *__restrict__ cams, int b) {
    const float *Kr = cams + (size_t)b * CAM_FLOATS, *K = Kr + 9, *E = Kr + 18;
    UnprojCam c;
    c.offset = Kr[2] - K[2]; c.tf = Kr[30]; c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.Rt[i * 3 + j] = E[j * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) c.Rtt[i] = c.Rt[i * 3] * E[3] + c.Rt[i * 3 + 1] * E[7] + c.Rt[i * 3 + 2] * E[11];
    return c;
}

bool fill(UnprojArgs &a, int B, int S, const float *ref_intr, const float *intr, const float *extr, const float *tf) {
    if (B < 0 || B > MAXB || S < 0) return false;
    a.B = B; a.S = S;
    for (int b = 0; b < B; b++) {
        const float *K = intr + 9 * b, *Kr = ref_intr + 9 * b, *E = extr + 12 * b;  // extr [B,3,4] (or the top of [B,4,4]: caller passes 3x4 rows)
        UnprojCam &c = a.cam[b];
        c.offset = Kr[2] - K[2]; c.tf = tf[b]; c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) c.Rt[i * 3 + j] = E[j * 4 + i];
        for (int i = 0; i < 3; i++) c.Rtt[i] = c.Rt[i * 3] * E[3] + c.Rt[i * 3 + 1] * E[7] + c.Rt[i * 3 + 2] * E[11];
    }
    return true;
}

}  // namespace

// Camera parameters are HOST pointers (a few dozen floats per sample): ref_intr, intr [B,3,3]; extr [B,3,4] row-major; Tf_x [B].
extern "C" int up_unproject_forward(int B, int S, const float *flow, const float *mask, int64_t mask_batch_stride, const float *ref_intr_host,
                                    const float *intr_host, const float *extr_host, const float *tf_host, float *depth, float *xyz,
                                    uint8_t *valid, void *stream) {
    if (B < 0 || S < 0 || !ref_intr_host || !intr_host || !extr_host || !tf_host) return GPSGS_E_INVALID;
    if (B == 0 || S == 0) return GPSGS_OK;
    if (!flow || !mask || !depth || !xyz || !valid) return GPSGS_E_INVALID;
    const size_t S2 = (size_t)S * S;
    for (int b0 = 0; b0 < B; b0 += MAXB) {  // the cameras travel by value in the kernel arguments, MAXB at a time: any batch size works
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        UnprojArgs a;
        if (!fill(a, nb, S, ref_intr_host + 9 * b0, intr_host + 9 * b0, extr_host + 12 * b0, tf_host + b0)) return GPSGS_E_INVALID;
        hipLaunchKernelGGL(k_unproject_fwd, dim3((S * S + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, a, (const float *)nullptr, flow + b0 * S2,
                           mask + (int64_t)b0 * mask_batch_stride, mask_batch_stride, depth + b0 * S2, xyz + b0 * S2 * 3, valid + b0 * S2);
    }
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int up_unproject_backward(int B, int S, const float *depth, const float *mask, int64_t mask_batch_stride, const float *ref_intr_host,
                                     const float *intr_host, const float *extr_host, const float *tf_host, const float *g_depth,
                                     const float *g_xyz, int64_t gx_batch_stride, int64_t gx_pixel_stride, int64_t gx_channel_stride,
                                     float *d_flow, void *stream) {
    if (B < 0 || S < 0 || !ref_intr_host || !intr_host || !extr_host || !tf_host) return GPSGS_E_INVALID;
    if (B == 0 || S == 0) return GPSGS_OK;
    if (!depth || !mask || !d_flow) return GPSGS_E_INVALID;
    const size_t S2 = (size_t)S * S;
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        UnprojArgs a;
        if (!fill(a, nb, S, ref_intr_host + 9 * b0, intr_host + 9 * b0, extr_host + 12 * b0, tf_host + b0)) return GPSGS_E_INVALID;
        hipLaunchKernelGGL(k_unproject_bwd, dim3((S * S + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, a, (const float *)nullptr, depth + b0 * S2,
                           mask + (int64_t)b0 * mask_batch_stride, mask_batch_stride, g_depth ? g_depth + b0 * S2 : nullptr,
                           g_xyz ? g_xyz + (int64_t)b0 * gx_batch_stride : nullptr, gx_batch_stride, gx_pixel_stride, gx_channel_stride,
                           d_flow + b0 * S2);
    }
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

// The same two entry points with the cameras in DEVICE memory: cams_dev[B][31] = {ref_intr 3x3, intr 3x3, extr rows 0..2 (3x4, row-major), Tf_x} per sample.
// Nothing is read on the host; any batch size in one launch.
extern "C" int up_unproject_forward_dev(int B, int S, const float *flow, const float *mask, int64_t mask_batch_stride, const float *cams_dev, float *depth,
                                        float *xyz, uint8_t *valid, void *stream) {
    if (B < 0 || S < 0) return GPSGS_E_INVALID;
    if (B == 0 || S == 0) return GPSGS_OK;
    if (!flow || !mask || !depth || !xyz || !valid || !cams_dev || B > 65535) return GPSGS_E_INVALID;
    UnprojArgs a;
    a.B = B; a.S = S;
    hipLaunchKernelGGL(k_unproject_fwd, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a, cams_dev, flow, mask, mask_batch_stride, depth, xyz, valid);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int up_unproject_backward_dev(int B, int S, const float *depth, const float *mask, int64_t mask_batch_stride, const float *cams_dev, const float *g_depth,
                                         const float *g_xyz, int64_t gx_batch_stride, int64_t gx_pixel_stride, int64_t gx_channel_stride, float *d_flow,
                                         void *stream) {
    if (B < 0 || S < 0) return GPSGS_E_INVALID;
    if (B == 0 || S == 0) return GPSGS_OK;
    if (!depth || !mask || !d_flow || !cams_dev || B > 65535) return GPSGS_E_INVALID;
    UnprojArgs a;
    a.B = B; a.S = S;
    hipLaunchKernelGGL(k_unproject_bwd, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a, cams_dev, depth, mask, mask_batch_stride, g_depth, g_xyz,
                       gx_batch_stride, gx_pixel_stride, gx_channel_stride, d_flow);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
