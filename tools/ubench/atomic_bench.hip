// Micro-benchmark: device-scope atomic throughput patterns on MI355X (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: one lane per wave issues K atomics, addresses = hash(wave,k) % naddr, stride `pad` u32
// mode 1: all 64 lanes issue K atomics to distinct addresses (lane-consecutive, stride pad)
// mode 2: like 0 but returning atomics (result consumed)
// mode 3: plain stores (reference)
__global__ void k(unsigned *buf, int K, unsigned naddr, int pad, int mode, unsigned *sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (int i = 0; i < K; i++) {
        unsigned h = (wave * 2654435761u + i * 40503u);
        if (mode == 0) { if (lane == 0) atomicAdd(&buf[(size_t)(h % naddr) * pad], 1u); }
        else if (mode == 1) { atomicAdd(&buf[(size_t)((h + lane) % naddr) * pad], 1u); }
        else if (mode == 2) { if (lane == 0) acc += atomicAdd(&buf[(size_t)(h % naddr) * pad], 1u); }
        else { if (lane == 0) buf[(size_t)(h % naddr) * pad] = i; }
    }
    if (acc == 0xdeadbeef) sink[0] = acc;
}

int main() {
    const size_t N = 64u << 20;  // 256 MB of u32
    unsigned *buf, *sink;
    CK(hipMalloc(&buf, N * 4)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 0, N * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int waves = 9376, K = 40;
    struct Cfg { int mode; unsigned naddr; int pad; const char *name; };
    std::vector<Cfg> cfgs = {
        {0, 5169, 1, "1-lane nonret, 5169 addr dense"}, {0, 5169, 32, "1-lane nonret, 5169 addr padded 128B"},
        {0, 1u << 20, 1, "1-lane nonret, 1M addr dense"}, {2, 5169, 32, "1-lane returning, 5169 addr padded"},
        {2, 1u << 20, 1, "1-lane returning, 1M addr"}, {1, 5169, 32, "64-lane nonret, 5169 addr padded"},
        {1, 1u << 20, 1, "64-lane nonret, 1M addr dense"}, {3, 5169, 32, "1-lane plain store, 5169 padded"},
    };
    for (auto &c : cfgs) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, K, c.naddr, c.pad, c.mode, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) {
                double ops = (double)waves * K * (c.mode == 1 ? 64 : 1);
                printf("%-42s %8.1f us  %8.2f Mops/ms  (%.2f ns/op chip-wide)\n", c.name, ms * 1e3, ops / ms / 1e3, ms * 1e6 / ops);
            }
        }
    }
    return 0;
}
