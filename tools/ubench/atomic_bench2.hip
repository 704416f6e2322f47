// Micro-benchmark 2: same-address atomic bursts from neighbouring waves (the binning pattern).  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 & 0x3f) | (0 << 6) | ((4 - 1) << 11)) & 0xf; }  // HW_REG_XCC_ID = 20

// each wave: K single-lane atomics to bins base(w)+i, base(w) = (w / share) * step  -> `share` consecutive waves hit the same bins
__global__ void k(unsigned *buf, int K, int share, int step, int pad, int mode, unsigned region, unsigned *sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    const unsigned base = (wave / share) * step;
    unsigned *b = buf;
    if (mode == 2) b = buf + (size_t)xcc_id() * region;          // XCD-private replica
    if (mode == 3) b = buf + (size_t)(blockIdx.x & 7) * region;  // replica by blockIdx%8
    for (int i = 0; i < K; i++) {
        unsigned *a = &b[(size_t)(base + i) * pad];
        if (lane == 0) {
            if (mode == 1) acc += atomicAdd(a, 1u);
            else if (mode == 4) __hip_atomic_fetch_add(a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(a, 1u);
        }
    }
    if (acc == 0xdeadbeef) sink[0] = acc;
}

int main() {
    const size_t N = 64u << 20;
    unsigned *buf, *sink;
    CK(hipMalloc(&buf, N * 4)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 0, N * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int waves = 9376, K = 11;
    struct Cfg { int share, step, pad, mode; const char *name; };
    std::vector<Cfg> cfgs = {
        {1, 11, 32, 0, "no sharing (each wave own bins)"}, {4, 4, 32, 0, "4 waves share"}, {16, 8, 32, 0, "16 waves share"},
        {64, 8, 32, 0, "64 waves share"}, {64, 8, 1, 0, "64 waves share, dense counters"}, {64, 8, 32, 1, "64 share, returning"},
        {64, 8, 32, 2, "64 share, XCC_ID-private replicas"}, {64, 8, 32, 3, "64 share, blockIdx%8 replicas"},
        {64, 8, 32, 4, "64 share, workgroup-scope atomics (INCORRECT across XCDs, timing only)"},
        {256, 8, 32, 0, "256 waves share"},
    };
    for (auto &c : cfgs) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, K, c.share, c.step, c.pad, c.mode, 1u << 22, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) best = ms < best ? ms : best;
        }
        printf("%-75s %8.1f us\n", c.name, best * 1e3);
    }
    return 0;
}
