// corr_sampler.hip -- 1-D correlation lookup for RAFT-Stereo on gfx950 (replaces the `corr_sampler` CUDA extension
// called at /root/reference/core/corr.py:22,28; semantics = CorrBlock1D, /root/reference/core/corr.py:127-146, i.e.
// linear interpolation of the 2r+1 taps x0-r..x0+r along W2 with zero padding).
//
// Forward: one thread per (n,y,x): 2r+2 contiguous reads of its volume row, 2r+1 stores coalesced along x.
// Backward: grad_volume row (n,y,x,:) is owned by exactly one (n,y,x), so there is no scatter and no atomic: one
// thread per grad_volume ELEMENT writes either zero or the sum of its (at most two) taps -- stores are fully
// coalesced along W2 and the zero-fill is fused (no separate memset).
#include <hip/hip_fp16.h>

#include "gsr_common.h"

#pragma clang fp contract(off)

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<__half>(const __half *p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<__half>(__half *p, float v) { *p = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(256) void k_cs_fwd(const T *__restrict__ volume, const float *__restrict__ coords, T *__restrict__ out,
                                                int total, int H1, int W1, int W2, int r) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // (n*H1 + y)*W1 + x
    if (idx >= total) return;
    const int hw = H1 * W1;
    const int n = idx / hw, yx = idx - n * hw;
    const float x0 = coords[idx];
    const float fl = floorf(x0);
    const float dx = x0 - fl;
    const int xs = (int)fl - r;
    const T *v = volume + (size_t)idx * W2;
    const int rd = 2 * r + 1;
    float prev = (xs >= 0 && xs < W2) ? ldf(v + xs) : 0.f;
    T *o = out + (size_t)n * rd * hw + yx;
    for (int k = 0; k < rd; k++) {
        const int x1 = xs + k + 1;
        const float next = (x1 >= 0 && x1 < W2) ? ldf(v + x1) : 0.f;
        stf(o + (size_t)k * hw, prev * (1.0f - dx) + next * dx);
        prev = next;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_cs_bwd(const float *__restrict__ coords, const T *__restrict__ grad_out,
                                                T *__restrict__ grad_volume, size_t total, int H1, int W1, int W2, int r) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((n*H1 + y)*W1 + x)*W2 + x1
    if (e >= total) return;
    const int x1 = (int)(e % W2);
    const size_t idx = e / W2;
    const int hw = H1 * W1;
    const int n = (int)(idx / hw), yx = (int)(idx - (size_t)n * hw);
    const float x0 = coords[idx];
    const float fl = floorf(x0);
    const float dx = x0 - fl;
    const int rd = 2 * r + 1;
    const int i = x1 - ((int)fl - r);  // tap index of this element, valid 0..rd
    float g = 0.f;
    if (i >= 0 && i <= rd) {
        const T *go = grad_out + (size_t)n * rd * hw + yx;
        if (i > 0) g += ldf(go + (size_t)(i - 1) * hw) * dx;
        if (i < rd) g += ldf(go + (size_t)i * hw) * (1.0f - dx);
    }
    stf(grad_volume + e, g);
}

// Same result, four consecutive x1 per thread (W2 % 4 == 0: every pyramid level of the reference's shapes): one coords load, at most five
// grad_out loads and ONE 16-byte (fp32) / 8-byte (fp16) store per thread instead of four of each -- the kernel is a pure store stream
// (grad_volume is 3.6x the bytes of everything it reads).  Per element the same two products are added in the same order as in k_cs_bwd.
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<__half> { typedef uint2 type; };
__device__ __forceinline__ float4 pack4(const float (&g)[4], float) { return make_float4(g[0], g[1], g[2], g[3]); }
__device__ __forceinline__ uint2 pack4(const float (&g)[4], __half) {
    const __half2 a = __halves2half2(__float2half(g[0]), __float2half(g[1])), b = __halves2half2(__float2half(g[2]), __float2half(g[3]));
    return make_uint2(*reinterpret_cast<const uint32_t *>(&a), *reinterpret_cast<const uint32_t *>(&b));
}
template <typename T>
__global__ __launch_bounds__(256) void k_cs_bwd4(const float *__restrict__ coords, const T *__restrict__ grad_out,
                                                 T *__restrict__ grad_volume, size_t total4, int H1, int W1, int W2, int r) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // quad index: (((n*H1 + y)*W1 + x)*W2 + x1) / 4
    if (q >= total4) return;
    const int wq = W2 >> 2;
    const int x1 = (int)(q % wq) * 4;
    const size_t idx = q / wq;
    const int hw = H1 * W1;
    const int n = (int)(idx / hw), yx = (int)(idx - (size_t)n * hw);
    const float x0 = coords[idx];
    const float fl = floorf(x0);
    const float dx = x0 - fl;
    const int rd = 2 * r + 1;
    const int i0 = x1 - ((int)fl - r);  // tap index of the first element
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (i0 + 3 >= 0 && i0 <= rd) {
        const T *go = grad_out + (size_t)n * rd * hw + yx;
        float t[5];  // grad_out taps i0-1 .. i0+3 (zero outside 0..rd-1: those terms are skipped below, as in k_cs_bwd)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = i0 - 1 + k;
            t[k] = (i >= 0 && i < rd) ? ldf(go + (size_t)i * hw) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k;
            float v = 0.f;
            if (i >= 0 && i <= rd) {
                if (i > 0) v += t[k] * dx;
                if (i < rd) v += t[k + 1] * (1.0f - dx);
            }
            g[k] = v;
        }
    }
    *reinterpret_cast<typename Vec4<T>::type *>(grad_volume + q * 4) = pack4(g, T());
}

}  // namespace

extern "C" int cs_forward(const void *volume, const float *coords, void *out, int N, int H1, int W1, int W2, int radius, int dtype,
                          void *stream) {
    if (N < 0 || H1 < 0 || W1 < 0 || W2 < 0 || radius < 0 || (dtype != 0 && dtype != 1)) return GPSGS_E_INVALID;
    const long long total = (long long)N * H1 * W1;
    if (total == 0) return GPSGS_OK;
    if (!volume || !coords || !out || total > 0x7fffffffLL) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_cs_fwd<float>, grid, dim3(256), 0, s, (const float *)volume, coords, (float *)out, (int)total, H1, W1, W2, radius);
    else
        hipLaunchKernelGGL(k_cs_fwd<__half>, grid, dim3(256), 0, s, (const __half *)volume, coords, (__half *)out, (int)total, H1, W1, W2, radius);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int cs_backward(const float *coords, const void *grad_out, void *grad_volume, int N, int H1, int W1, int W2, int radius,
                           int dtype, void *stream) {
    if (N < 0 || H1 < 0 || W1 < 0 || W2 < 0 || radius < 0 || (dtype != 0 && dtype != 1)) return GPSGS_E_INVALID;
    const size_t total = (size_t)N * H1 * W1 * W2;
    if (total == 0) return GPSGS_OK;
    if (!coords || !grad_out || !grad_volume) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if ((W2 & 3) == 0 && ((uintptr_t)grad_volume & 15) == 0) {
        const size_t total4 = total / 4;
        const dim3 grid4((unsigned)((total4 + 255) / 256));
        if (dtype == 0)
            hipLaunchKernelGGL(k_cs_bwd4<float>, grid4, dim3(256), 0, s, coords, (const float *)grad_out, (float *)grad_volume, total4, H1, W1, W2, radius);
        else
            hipLaunchKernelGGL(k_cs_bwd4<__half>, grid4, dim3(256), 0, s, coords, (const __half *)grad_out, (__half *)grad_volume, total4, H1, W1, W2, radius);
        return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
    }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_cs_bwd<float>, grid, dim3(256), 0, s, coords, (const float *)grad_out, (float *)grad_volume, total, H1, W1, W2, radius);
    else
        hipLaunchKernelGGL(k_cs_bwd<__half>, grid, dim3(256), 0, s, coords, (const __half *)grad_out, (__half *)grad_volume, total, H1, W1, W2, radius);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
