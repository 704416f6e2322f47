// Micro-benchmark 3: the backward flush pattern -- every lane adds 9 floats into the record of a (random) Gaussian.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// mode 0: AoS 12-float records, 9 atomics per lane;  1: SoA 9 arrays;  2: AoS but plain stores (reference);
// mode 3: AoS 16-float (64 B) records;  4: SoA with 64-bit double atomics on pairs (5 atomics)
__global__ void k(float *buf, const unsigned *ids, int n, int P, int mode) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned id = ids[t];
    const float v = 1.0f + (t & 7);
    if (mode == 0) { float *d = buf + (size_t)id * 12; for (int k = 0; k < 9; k++) atomicAdd(d + k, v); }
    else if (mode == 1) { for (int k = 0; k < 9; k++) atomicAdd(buf + (size_t)k * P + id, v); }
    else if (mode == 2) { float *d = buf + (size_t)id * 12; for (int k = 0; k < 9; k++) d[k] = v; }
    else if (mode == 3) { float *d = buf + (size_t)id * 16; for (int k = 0; k < 9; k++) atomicAdd(d + k, v); }
    else if (mode == 4) { double *d = (double *)buf; for (int k = 0; k < 5; k++) atomicAdd(d + (size_t)k * P + id, (double)v); }
}
int main() {
    const int P = 600000, n = 1850000;
    float *buf; unsigned *ids;
    CK(hipMalloc(&buf, (size_t)P * 16 * 8)); CK(hipMemset(buf, 0, (size_t)P * 16 * 8));
    std::vector<unsigned> h(n);
    // bin-like locality: consecutive lanes take nearby (not identical) Gaussians, each Gaussian appears ~3x overall
    unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (unsigned)(((long long)(i / 64) * 21 + (s >> 8) % 1400) % P); }
    CK(hipMalloc(&ids, n * 4)); CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[] = {"AoS 48B records, 9 f32 atomics/lane", "SoA 9 arrays, 9 f32 atomics/lane", "AoS plain stores (reference)", "AoS 64B records, 9 f32 atomics/lane", "SoA f64 atomics x5"};
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, buf, ids, n, P, mode);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) best = ms < best ? ms : best;
        }
        printf("%-40s %8.1f us\n", names[mode], best * 1e3);
    }
    return 0;
}
