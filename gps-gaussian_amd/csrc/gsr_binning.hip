// gsr_binning.hip -- tile binning and per-tile depth ordering for gfx950.
//
// Upstream builds one global list of (tile<<32 | depth) keys and radix-sorts all R of them through HBM (~6 passes
// of 24 B/instance) after a blocking D2H read of R.  Here the tile part of the key is
// resolved by construction instead of by sorting:
//   k_scan     exclusive scan of the per-tile counts (written by k_preprocess) -> tile_offset, R, overflow flag;
//              R never leaves the device.
//   k_scatter  each Gaussian drops (depth_bits<<32 | id) into its tiles' segments (one returning atomic per
//              instance on a per-tile cursor).
//   k_sort_*   one workgroup per tile sorts its segment in LDS with an ascending-only bitonic network on the
//              64-bit key.  Keys are unique (id in the low word) so the result is deterministic and equals
//              upstream's stable radix order: depth ascending, ties by Gaussian index.
//              Segments longer than the LDS capacity fall back to the same network run in global memory.
// HBM traffic: 8 B written + 8 B read + 4 B written per instance, versus >= 144 B for the global radix sort.
#include "gsr_common.h"

namespace {

constexpr int SCAN_THREADS = 1024;

// Single-workgroup scan: T is at most a few 10^4 tiles (16,384 at 2048^2), i.e. a few microseconds.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_offset,
                                                       uint32_t *__restrict__ tile_cursor, int T, int64_t cap,
                                                       GsrHeader *__restrict__ hdr) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ uint32_t wmax[SCAN_THREADS / 64];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    uint32_t vmax = 0;
    __syncthreads();
    for (int base = 0; base < T; base += SCAN_THREADS) {
        const int i = base + tid;
        const uint32_t v = i < T ? tile_count[i] : 0u;
        vmax = v > vmax ? v : vmax;
        uint32_t x = v;  // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const uint32_t carry = carry_s;
        const uint32_t excl = carry + woff + x - v;
        if (i < T) {
            tile_offset[i] = excl;
            tile_cursor[i] = excl;
        }
        __syncthreads();
        if (tid == SCAN_THREADS - 1) carry_s = excl + v;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t y = __shfl_xor(vmax, d, 64);
        vmax = y > vmax ? y : vmax;
    }
    if (lane == 0) wmax[wid] = vmax;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < SCAN_THREADS / 64; w++) m = wmax[w] > m ? wmax[w] : m;
        const uint32_t R = carry_s;
        tile_offset[T] = R;
        hdr->num_rendered = R;
        hdr->overflow = ((int64_t)R > cap) ? 1u : 0u;
        hdr->max_tile_count = m;
    }
}

__global__ __launch_bounds__(256) void k_scatter(int P, int gx, const GsrSplat *__restrict__ splats, uint32_t *__restrict__ tile_cursor,
                                                 uint64_t *__restrict__ keys, const GsrHeader *__restrict__ hdr) {
    if (hdr->overflow) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 c = reinterpret_cast<const float4 *>(splats + i)[2];
    const uint32_t lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
    const int r0x = lo & 0xffff, r0y = lo >> 16, r1x = hi & 0xffff, r1y = hi >> 16;
    if (r1x <= r0x || r1y <= r0y) return;
    const uint64_t key = ((uint64_t)__float_as_uint(c.y) << 32) | (uint32_t)i;
    for (int ty = r0y; ty < r1y; ty++)
        for (int tx = r0x; tx < r1x; tx++) {
            const uint32_t pos = atomicAdd(&tile_cursor[ty * gx + tx], 1u);
            keys[pos] = key;
        }
}

// ascending compare-exchange on LDS / global arrays, virtual +inf padding beyond n (comparators with j >= n are no-ops)
template <typename PtrT>
__device__ __forceinline__ void cmpx(PtrT a, uint32_t i, uint32_t j, uint32_t n) {
    if (j < n) {
        const uint64_t x = a[i], y = a[j];
        if (x > y) {
            a[i] = y;
            a[j] = x;
        }
    }
}

// Bitonic network with ascending-only comparators ("flip" first step, then half-cleaners).
template <int THREADS, typename PtrT>
__device__ __forceinline__ void bitonic_sort(PtrT a, uint32_t n, uint32_t n2, int tid) {
    const uint32_t half = n2 >> 1;
    for (uint32_t kb = 1; (1u << kb) <= n2; kb++) {  // k = 2^kb: merge sorted runs of k/2 into runs of k
        const uint32_t hk = 1u << (kb - 1);
        for (uint32_t idx = tid; idx < half; idx += THREADS) {
            const uint32_t blk = idx >> (kb - 1), off = idx & (hk - 1);
            cmpx(a, (blk << kb) + off, (blk << kb) + (2 * hk - 1) - off, n);
        }
        __syncthreads();
        for (int jb = (int)kb - 2; jb >= 0; jb--) {  // half-cleaners, stride j = 2^jb
            const uint32_t j = 1u << jb;
            for (uint32_t idx = tid; idx < half; idx += THREADS) {
                const uint32_t i = ((idx >> jb) << (jb + 1)) | (idx & (j - 1));
                cmpx(a, i, i + j, n);
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// One workgroup per tile; handles segments with lo < n <= CAP in LDS; if GLOBAL_FALLBACK also n > CAP in HBM.
template <int THREADS, int CAP, bool GLOBAL_FALLBACK>
__global__ __launch_bounds__(THREADS) void k_sort(const uint32_t *__restrict__ tile_offset, uint64_t *__restrict__ keys,
                                                  uint32_t *__restrict__ point_list, uint32_t lo, const GsrHeader *__restrict__ hdr) {
    __shared__ uint64_t sk[CAP];
    if (hdr->overflow) return;
    const int t = blockIdx.x, tid = threadIdx.x;
    const uint32_t off = tile_offset[t], n = tile_offset[t + 1] - off;
    if (n <= lo) return;
    if (n > (uint32_t)CAP && !GLOBAL_FALLBACK) return;
    uint64_t *seg = keys + off;
    if (n == 1) {
        if (tid == 0) point_list[off] = (uint32_t)seg[0];
        return;
    }
    const uint32_t n2 = next_pow2(n);
    if (n <= (uint32_t)CAP) {
        for (uint32_t i = tid; i < n; i += THREADS) sk[i] = seg[i];
        __syncthreads();
        bitonic_sort<THREADS>(sk, n, n2, tid);
        for (uint32_t i = tid; i < n; i += THREADS) point_list[off + i] = (uint32_t)sk[i];
    } else {
        // rare: tile list longer than LDS; same network in global memory (workgroup-coherent through L2/L1 of this CU)
        volatile uint64_t *vs = seg;
        bitonic_sort<THREADS>(vs, n, n2, tid);
        for (uint32_t i = tid; i < n; i += THREADS) point_list[off + i] = (uint32_t)vs[i];
    }
}

}  // namespace

void gsr_launch_scan(uint32_t *tile_count, uint32_t *tile_offset, uint32_t *tile_cursor, int T, int64_t cap, GsrHeader *hdr, hipStream_t s) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_THREADS), 0, s, tile_count, tile_offset, tile_cursor, T, cap, hdr);
}

void gsr_launch_scatter(int P, int gx, const GsrSplat *splats, uint32_t *tile_cursor, uint64_t *keys, int64_t cap, const GsrHeader *hdr,
                        hipStream_t s) {
    (void)cap;
    if (P <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, splats, tile_cursor, keys, hdr);
}

void gsr_launch_sort(int T, const uint32_t *tile_offset, uint64_t *keys, uint32_t *point_list, const GsrHeader *hdr, hipStream_t s) {
    if (T <= 0) return;
    // two size classes so that typical tiles keep several workgroups per CU: <= 2048 keys (16 KiB LDS) and
    // <= 16384 keys (128 KiB LDS, 1024 threads); anything longer sorts in global memory inside the second kernel.
    hipLaunchKernelGGL((k_sort<256, 2048, false>), dim3(T), dim3(256), 0, s, tile_offset, keys, point_list, 0u, hdr);
    hipLaunchKernelGGL((k_sort<1024, 16384, true>), dim3(T), dim3(1024), 0, s, tile_offset, keys, point_list, 2048u, hdr);
}
