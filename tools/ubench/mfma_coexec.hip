// Micro-benchmark: does matrix-core work overlap with vector-ALU work on one SIMD of gfx950?
// Each wave runs ITER iterations of a body: mode bit 0 = NV independent v_fma_f32 chains' worth of VALU work, bit 1 = one MFMA chain step.
// variants: f32-input MFMA (v_mfma_f32_32x32x2_f32, 64 cycles) and bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32 cycles).
// Compare t(VALU only), t(MFMA only), t(both): overlap <=> t(both) ~ max, no overlap <=> t(both) ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int KIND, int NV>
__global__ __launch_bounds__(64) void k(float *out, int iters) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (float)threadIdx.x * 0.001f + i;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float a = 1.0001f, b = 0.5f;
    bf16x8 pa, pb;
#pragma unroll
    for (int i = 0; i < 8; i++) { pa[i] = (short)(0x3f80 + i); pb[i] = (short)(0x3f00 + threadIdx.x); }
    for (int it = 0; it < iters; it++) {
        if (MODE & 2) {
            if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
        }
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < NV; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE, int KIND, int NV>
float run(float *out, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, KIND, NV>), dim3(blocks), dim3(64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

int main() {
    float *out; CK(hipMalloc(&out, 1 << 24));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {      // waves per SIMD
        const int blocks = 256 * 4 * wps;        // one wave per block, 256 CUs x 4 SIMDs
        printf("waves/SIMD %d (VALU body = 8*NV v_fma per iteration)\n", wps);
        printf("  f32 MFMA 32x32x2 : NV=2  valu %7.0f  mfma %7.0f  both %7.0f us\n", run<1, 0, 2>(out, blocks, iters), run<2, 0, 2>(out, blocks, iters), run<3, 0, 2>(out, blocks, iters));
        printf("  bf16 MFMA 32x32x16: NV=1  valu %7.0f  mfma %7.0f  both %7.0f us\n", run<1, 1, 1>(out, blocks, iters), run<2, 1, 1>(out, blocks, iters), run<3, 1, 1>(out, blocks, iters));
        printf("  bf16 MFMA 32x32x16: NV=2  valu %7.0f  mfma %7.0f  both %7.0f us\n", run<1, 1, 2>(out, blocks, iters), run<2, 1, 2>(out, blocks, iters), run<3, 1, 2>(out, blocks, iters));
        printf("  bf16 MFMA 32x32x16: NV=4  valu %7.0f  mfma %7.0f  both %7.0f us\n", run<1, 1, 4>(out, blocks, iters), run<2, 1, 4>(out, blocks, iters), run<3, 1, 4>(out, blocks, iters));
    }
    return 0;
}
