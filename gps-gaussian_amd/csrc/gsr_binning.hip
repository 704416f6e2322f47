// gsr_binning.hip -- bin assignment and per-bin depth ordering for gfx950.
//
// Upstream builds one global list of (tile<<32 | depth) keys and radix-sorts all R of them through HBM (~6 passes
// of 24 B/instance) after a blocking D2H read of R.  Here the bin part of the key is
// resolved by construction instead of by sorting:
//   k_scan_a/b two-phase parallel exclusive scan of the per-bin counts written by k_preprocess (normally fused into ONE launch:
//              release flags + wait) -> bin_offset, cursors, R, overflow flag; R is published to the host from here.  The same pass
//              emits wg_order: compositing workgroups with work first (so every CU starts on real work and the empty
//              ones drain in the gaps), in image order within each class (keeps neighbouring bins on neighbouring CUs).
//   k_scatter  each Gaussian drops (depth_bits<<32 | id) into its bins' segments; slots are reserved with ONE returning
//              global atomic per (workgroup, bin), not one per instance; the per-workgroup bin table and the per-cell hit
//              masks are the ones k_preprocess recorded (gsr_block_emit), nothing is recounted or re-tested.
//   k_sort_*   one WAVE per bin sorts its segment with an ascending-only bitonic network on the 64-bit key, keys in
//              registers, exchanges by DPP / LDS crossbar (lists <= 1024); longer lists get a 1024-thread workgroup in LDS.  Keys are unique (id in the low word) so the result is deterministic and equals
//              upstream's stable radix order: depth ascending, ties by Gaussian index.
//              Segments longer than the LDS capacity fall back to the same network run in global memory.
// HBM traffic: 8 B written + 8 B read + 4 B written per instance, versus >= 144 B for the global radix sort.
#include "gsr_common.h"

namespace {

constexpr int SB = GSR_SCAN_BLOCK;  // 1024 bins per scan block

// Work order of the compositing / sorting waves: the bin grid enumerated in PATCHES of 8x8 bins (64x64 pixels; row-major inside a
// patch, patches row-major).  A Gaussian is listed in ~3 neighbouring bins, mostly a 2x2 block: in this order the block's bins are
// (with probability ~0.77) within 64 consecutive entries, which xcd_list_pos() hands to ONE XCD -- one L2 then fetches the splat
// record once for all of them and merges the gradient records the compositing backward scatters into the Gaussian's (contiguous)
// slots.  Image order (runs of 64x1 bins) shared only the horizontal neighbours.  -1: the index has no bin (ragged grid edge).
__device__ __forceinline__ int tiled_bin(uint32_t t, int bx, int by) {
    const uint32_t pgx = ((uint32_t)bx + 7u) >> 3;
    const uint32_t p = t >> 6, w = t & 63u;
    const uint32_t x = (p % pgx) * 8u + (w & 7u), y = (p / pgx) * 8u + (w >> 3);
    return (x < (uint32_t)bx && y < (uint32_t)by) ? (int)(y * (uint32_t)bx + x) : -1;
}

// block-wide exclusive scan of one uint per thread (1024 threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *wsum /*[16]*/, uint32_t *total) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();  // wsum may still be read from a previous call
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SB / 64; w++) {
        const uint32_t s = wsum[w];
        if (w < wid) woff += s;
        tot += s;
    }
    *total = tot;
    return woff + x - v;
}

// Work classes of a bin in the compositing / sort order, by the length of its list relative to the longest list an earlier, similar
// view produced (hint: GsrViewExt.order_hint, a kernel argument; 0 = unknown = every busy bin in class 0 = plain patch order): class 0 = more than 1/2 of it, 1 = more than 1/4, 2 = any other busy bin, 3 = idle.  The classes are
// dispatched in that order: a bin is ONE wave's sequential job and a SIMD gets only ~5 of them per kernel, so a 900-entry list that
// starts late is what the other SIMDs end up waiting for (longest-processing-time-first, coarsely).  Two packed counters:
// w0 = class 0 | class 1 << 16, w1 = class 2 | idle << 16 (each count <= 1024 per scan block).
// The order is a performance hint only (every bin is handled exactly once whatever its position), so the threshold may be anything:
// the caller passes what the header of an earlier forward reported (max_tile_count).
__device__ __forceinline__ int work_class(int wb, uint32_t wc, uint32_t hint) {
    return wb < 0 ? -1 : (wc > (hint >> 1) ? 0 : (wc > (hint >> 2) ? 1 : (wc > 0u ? 2 : 3)));
}
__device__ __forceinline__ uint32_t class_w0(int cls) { return cls == 0 ? 1u : (cls == 1 ? 0x10000u : 0u); }
__device__ __forceinline__ uint32_t class_w1(int cls) { return cls == 2 ? 1u : (cls == 3 ? 0x10000u : 0u); }

// phase A: per block of 1024 indices -> part[2 blk] = {sum of counts (bins in image order), w0 of the block's indices in WORK order
// (tiled_bin), max count, -}, part[2 blk + 1].x = their w1
__global__ __launch_bounds__(SB) void k_scan_a(const uint32_t *__restrict__ bin_count, uint4 *__restrict__ part, int NB, int bx, int by,
                                               uint32_t hint) {
    __shared__ uint32_t red[4][SB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x * SB + tid;
    const uint32_t c = b < NB ? bin_count[(size_t)b * GSR_CPAD] : 0u;
    const int wb = tiled_bin((uint32_t)b, bx, by);
    const uint32_t wc = wb >= 0 ? bin_count[(size_t)wb * GSR_CPAD] : 0u;
    const int cls = work_class(wb, wc, hint);
    uint32_t s = c, nb = class_w0(cls), ni = class_w1(cls), mx = c;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d, 64);
        nb += __shfl_xor(nb, d, 64);
        ni += __shfl_xor(ni, d, 64);
        const uint32_t y = __shfl_xor(mx, d, 64);
        mx = y > mx ? y : mx;
    }
    if (lane == 0) { red[0][wid] = s; red[1][wid] = nb; red[2][wid] = mx; red[3][wid] = ni; }
    __syncthreads();
    if (tid == 0) {
        uint32_t ts = 0, tb = 0, tm = 0, ti = 0;
        for (int w = 0; w < SB / 64; w++) { ts += red[0][w]; tb += red[1][w]; tm = red[2][w] > tm ? red[2][w] : tm; ti += red[3][w]; }
        part[2 * blockIdx.x] = make_uint4(ts, tb, tm, 0u);
        part[2 * blockIdx.x + 1] = make_uint4(ti, 0u, 0u, 0u);
    }
}

// phase B: every block re-derives its prefix from the (few) block partials, then scans its own 1024 bins.
// FUSED = true: phase A runs in the same launch -- every block publishes its partial with a release flag and waits for the
// flags of all blocks (the total of ALL blocks is needed, so all blocks must be co-resident: the launcher only fuses up to
// GSR_SCAN_FUSE_MAX blocks = 65,536 bins, a 2048^2 image, an eighth of the chip's 512 resident 1024-thread workgroups -- so
// even several forwards racing on different streams cannot starve each other's scan blocks; beyond that the two-launch
// form is used).  Saves one launch.
#define GSR_SCAN_FUSE_MAX 64
template <bool FUSED>
__global__ __launch_bounds__(SB) void k_scan_b(const uint32_t *__restrict__ bin_count, uint4 *__restrict__ part,
                                               uint32_t *__restrict__ bin_offset, uint32_t *__restrict__ bin_cursor,
                                               uint32_t *__restrict__ wg_order, int NB, int bx, int by, int nblocks, int64_t cap,
                                               GsrHeader *__restrict__ hdr, uint32_t *__restrict__ gpart, int n_gblocks,
                                               uint32_t *__restrict__ host_hdr, uint32_t host_seq, int no_large_sort,
                                               uint32_t hint) {
    __shared__ uint32_t wsum[SB / 64];
    __shared__ uint4 sp[FUSED ? GSR_SCAN_FUSE_MAX : 1];
    __shared__ uint32_t sp_idle[FUSED ? GSR_SCAN_FUSE_MAX : 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.x * SB + tid;
    const uint32_t c = b < NB ? bin_count[(size_t)b * GSR_CPAD] : 0u;
    const int wb = tiled_bin((uint32_t)b, bx, by);  // the bin this thread places in the work order
    const uint32_t wc = wb >= 0 ? bin_count[(size_t)wb * GSR_CPAD] : 0u;
    const int cls = work_class(wb, wc, hint);
    // block 0 also owns the per-Gaussian slot prefix (training) and the header.  Neither depends on the other blocks' bins, so the
    // slot scan runs first -- under the wait for their partials -- and the header leaves (also towards the host) as soon as the
    // totals are known, before this block scans its own bins.
    uint32_t tot_slots = 0;
    auto slot_scan = [&]() {
        if (blockIdx.x == 0 && gpart) {  // block sums of the per-Gaussian slot counts -> exclusive prefix, in place
            uint32_t carry = 0;
            for (int base = 0; base < n_gblocks; base += SB) {
                const int k = base + tid;
                const uint32_t v = k < n_gblocks ? gpart[k] : 0u;
                uint32_t tot;
                const uint32_t ex = block_exscan(v, wsum, &tot);
                if (k < n_gblocks) gpart[k] = carry + ex;
                carry += tot;
            }
            tot_slots = carry;  // the gradient records need one slot per bin-rect cell (>= R: exact culling only removes instances)
        }
    };
    if (!FUSED) slot_scan();
    if (FUSED) {
        __shared__ uint32_t red[4][SB / 64];
        const int lane = tid & 63, wid = tid >> 6;
        uint32_t s = c, nb = class_w0(cls), ni = class_w1(cls), mx = c;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s += __shfl_xor(s, d, 64);
            nb += __shfl_xor(nb, d, 64);
            ni += __shfl_xor(ni, d, 64);
            const uint32_t y = __shfl_xor(mx, d, 64);
            mx = y > mx ? y : mx;
        }
        if (lane == 0) { red[0][wid] = s; red[1][wid] = nb; red[2][wid] = mx; red[3][wid] = ni; }
        __syncthreads();
        if (tid == 0) {
            uint32_t ts = 0, tb = 0, tm = 0, ti = 0;
            for (int w = 0; w < SB / 64; w++) { ts += red[0][w]; tb += red[1][w]; tm = red[2][w] > tm ? red[2][w] : tm; ti += red[3][w]; }
            uint32_t *me = reinterpret_cast<uint32_t *>(part + 2 * blockIdx.x);  // {sum, w0, max, flag | w1, -, -, -}
            me[0] = ts; me[1] = tb; me[2] = tm; me[4] = ti;
            __hip_atomic_store(me + 3, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // ready flag (zeroed by the forward's memset)
        }
        slot_scan();  // after this block's partial is out, before the others' are needed
        if (tid < nblocks) {  // wait for every block's partial (all blocks are resident: nblocks <= GSR_SCAN_FUSE_MAX)
            uint32_t *other = reinterpret_cast<uint32_t *>(part + 2 * tid);
            while (__hip_atomic_load(other + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
            sp[tid] = make_uint4(__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(other + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(other + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 1u);
            sp_idle[tid] = __hip_atomic_load(other + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    uint32_t pre_sum = 0, pre[4] = {0, 0, 0, 0}, tot[3] = {0, 0, 0}, tot_max = 0;  // pre / tot: per work class (tot: the three busy ones)
    uint64_t tot_sum = 0;  // 64-bit: a total beyond 2^32 must read as an overflow, not wrap below the capacity
    for (int i = 0; i < nblocks; i++) {  // a handful of uint4
        const uint4 p = FUSED ? sp[i] : part[2 * i];
        const uint32_t w1 = FUSED ? sp_idle[i] : part[2 * i + 1].x;
        if (i < (int)blockIdx.x) { pre_sum += p.x; pre[0] += p.y & 0xffffu; pre[1] += p.y >> 16; pre[2] += w1 & 0xffffu; pre[3] += w1 >> 16; }
        tot_sum += p.x; tot[0] += p.y & 0xffffu; tot[1] += p.y >> 16; tot[2] += w1 & 0xffffu; tot_max = p.z > tot_max ? p.z : tot_max;
    }
    const uint32_t tot_busy = tot[0] + tot[1] + tot[2];
    if (blockIdx.x == 0 && tid == 0) {
        bin_offset[NB] = (uint32_t)tot_sum;
        hdr->num_rendered = tot_sum;
        // a list longer than 1024 entries while the caller skipped the large-list sort launch is reported like an overflow:
        // every later kernel exits, the caller sees max_tile_count > 1024 and calls again with that launch
        const bool ovf_b = (int64_t)tot_sum > cap || (int64_t)tot_slots > cap || (no_large_sort && tot_max > 1024u) || hdr->row_overflow != 0u;
        hdr->overflow = ovf_b ? 1u : 0u;
        hdr->max_tile_count = tot_max;
        hdr->num_busy_wgs = tot_busy;
        hdr->num_slots = tot_slots;
        if (host_hdr) {
            // early notification: the header goes straight to host-coherent pinned memory from here, so the host can check
            // capacity while scatter / sort / compositing are still running (no copy engine, no event in the stream)
            const uint32_t ovf = ovf_b ? 1u : 0u;
            volatile uint32_t *h = host_hdr;
            h[0] = (uint32_t)tot_sum; h[1] = (uint32_t)(tot_sum >> 32); h[2] = ovf; h[3] = tot_max; h[4] = tot_busy; h[5] = tot_slots; h[6] = hdr->num_points;
            __threadfence_system();
            __hip_atomic_store(host_hdr + 7, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // the host polls this word
        }
    }
    uint32_t blk_total;
    const uint32_t off = pre_sum + block_exscan(c, wsum, &blk_total);
    if (b < NB) {
        bin_offset[b] = off;
        bin_cursor[(size_t)b * GSR_CPAD] = off;
    }
    // work-ordered workgroup list: the classes in order (longest lists first, idle bins last), patch order (tiled_bin) inside each class
    uint32_t blk_tot;
    const uint32_t r0 = block_exscan(class_w0(cls), wsum, &blk_tot);
    const uint32_t r1 = block_exscan(class_w1(cls), wsum, &blk_tot);
    if (wb >= 0) {
        const uint32_t pos = cls == 0 ? pre[0] + (r0 & 0xffffu)
                           : cls == 1 ? tot[0] + pre[1] + (r0 >> 16)
                           : cls == 2 ? tot[0] + tot[1] + pre[2] + (r1 & 0xffffu)
                                      : tot_busy + pre[3] + (r1 >> 16);
        wg_order[pos] = (uint32_t)wb;
    }
}

__global__ __launch_bounds__(GSR_BIN_THREADS) void k_scatter(int P, const uint32_t *__restrict__ row_range, int bx, const GsrSplat *__restrict__ splats, const uint32_t *__restrict__ hitmask,
                                                            const uint32_t *__restrict__ wg_tab, uint32_t *__restrict__ bin_cursor, uint64_t *__restrict__ keys,
                                                            const GsrHeader *__restrict__ hdr, const uint32_t *__restrict__ goff,
                                                            const uint32_t *__restrict__ gpart, uint8_t *__restrict__ inst_valid) {
    if (hdr->overflow) return;
    const int i = blockIdx.x * GSR_BIN_THREADS + threadIdx.x;
    {
        uint32_t row0;
        gsr_view_rows(row_range, P, row0, P);  // with a row range P was only the capacity: records behind the view's last Gaussian were never written
    }
    if ((int)(blockIdx.x * GSR_BIN_THREADS) >= P) return;  // workgroup entirely behind the view's last Gaussian (before any barrier: uniform)
    if (inst_valid) {
        // training: "no gradient record yet" for every slot of this workgroup's 1024 Gaussians (replaces a cap-byte memset).  Their slots are ONE
        // contiguous run [gpart[blk], gpart[blk + 1]) -- cleared by the whole workgroup with 16-byte stores.  (Until round 4 every thread cleared
        // its own Gaussian's run byte by byte: fine for the ~4-cell rects of trained scales, 100 strided byte stores per thread with scales at
        // their clamp.)
        const uint32_t s_beg = gpart[blockIdx.x], s_end = (blockIdx.x + 1 < gridDim.x) ? gpart[blockIdx.x + 1] : hdr->num_slots;
        for (uint32_t k = (s_beg & ~15u) + (uint32_t)threadIdx.x * 16u; k < s_end; k += GSR_BIN_THREADS * 16u) {
            if (k >= s_beg && k + 16u <= s_end) {
                *reinterpret_cast<uint4 *>(inst_valid + k) = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (uint32_t b = (k > s_beg ? k : s_beg); b < k + 16u && b < s_end; b++) inst_valid[b] = 0;
            }
        }
    }
    uint32_t lo = 0, hi = 0, mask = 0;
    uint64_t key = 0;
    GsrHit hit = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f};
    if (i < P) {
        mask = hitmask[i];  // the exact ellipse/bin tests of this Gaussian's rect, done once by k_preprocess
        const float4 *rec = reinterpret_cast<const float4 *>(splats + i);
        const float4 c = rec[2];
        lo = __float_as_uint(c.z);
        hi = __float_as_uint(c.w);
        key = ((uint64_t)__float_as_uint(c.y) << 32) | (uint32_t)i;
        if ((hi & 0xffff) > (lo & 0xffff)) {  // listed somewhere
            const uint32_t area = ((hi & 0xffff) - (lo & 0xffff)) * ((hi >> 16) - (lo >> 16));
            if (area > 32u) {  // rect too large for the cached mask: the predicate k_preprocess counted with (row intervals, or per cell for
                               // ill-conditioned conics: gsr_masked_hit), re-evaluated from the record and the threshold k_preprocess left in the
                               // mask word (never recomputed here: gsr_hit_from_threshold)
                const float4 a = rec[0], b = rec[1];
                hit = gsr_hit_from_threshold(a.x, a.y, a.z, a.w, b.x, __uint_as_float(mask));
            }
        }
    }
    gsr_block_emit(
        wg_tab + (size_t)blockIdx.x * GSR_WG_TAB_WORDS, lo, hi, bx, gsr_masked_hit(hit, mask, lo, hi), [&](int bin, uint32_t cnt) { return atomicAdd(&bin_cursor[(size_t)bin * GSR_CPAD], cnt); },
        [&](uint32_t pos, uint32_t) { keys[pos] = key; });
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

__device__ __forceinline__ void sort_sync(bool wave_only) {
    if (wave_only) {  // a single wave owns the segment: the DS queue is in order, only the compiler needs pinning
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Bitonic network with ascending-only comparators ("flip" first step, then half-cleaners).
template <int THREADS, typename PtrT>
__device__ __forceinline__ void bitonic_sort(PtrT a, uint32_t n, uint32_t n2, int tid) {
    constexpr bool WV = THREADS == 64;
    const uint32_t half = n2 >> 1;
    for (uint32_t kb = 1; (1u << kb) <= n2; kb++) {  // k = 2^kb: merge sorted runs of k/2 into runs of k
        const uint32_t hk = 1u << (kb - 1);
        for (uint32_t idx = tid; idx < half; idx += THREADS) {
            const uint32_t blk = idx >> (kb - 1), off = idx & (hk - 1);
            cmpx(a, (blk << kb) + off, (blk << kb) + (2 * hk - 1) - off, n);
        }
        sort_sync(WV);
        for (int jb = (int)kb - 2; jb >= 0; jb--) {  // half-cleaners, stride j = 2^jb
            const uint32_t j = 1u << jb;
            for (uint32_t idx = tid; idx < half; idx += THREADS) {
                const uint32_t i = ((idx >> jb) << (jb + 1)) | (idx & (j - 1));
                cmpx(a, i, i + j, n);
            }
            sort_sync(WV);
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

template <int THREADS, int CAP, bool GLOBAL_FALLBACK>
__device__ __forceinline__ void sort_one_bin(uint64_t *sk, uint32_t off, uint32_t n, uint64_t *__restrict__ keys,
                                             uint32_t *__restrict__ point_list, int tid) {
    uint64_t *seg = keys + off;
    if (n == 1) {
        if (tid == 0) point_list[off] = (uint32_t)seg[0];
        return;
    }
    const uint32_t n2 = next_pow2(n);
    if (n <= (uint32_t)CAP) {
        for (uint32_t i = tid; i < n; i += THREADS) sk[i] = seg[i];
        sort_sync(THREADS == 64);
        bitonic_sort<THREADS>(sk, n, n2, tid);
        for (uint32_t i = tid; i < n; i += THREADS) point_list[off + i] = (uint32_t)sk[i];
    } else if (GLOBAL_FALLBACK) {
        // rare: bin list longer than LDS; same network in global memory (one workgroup, coherent through its CU)
        volatile uint64_t *vs = seg;
        bitonic_sort<THREADS>(vs, n, n2, tid);
        for (uint32_t i = tid; i < n; i += THREADS) point_list[off + i] = (uint32_t)vs[i];
    }
}

// ---- register-resident bitonic sort: one wave, KPL keys per lane (element e = lane*KPL + r), no LDS memory traffic ------
// The LDS network above moves 4 x 64-bit LDS accesses per compare-exchange and was LDS-throughput bound (45 steps x 4 CEs
// per lane x ~16 LDS cycles for 512 keys).  Here every partner is `e ^ mask`, so strides below KPL are register-to-
// register and the rest are lane-xor exchanges (ds_bpermute through the LDS crossbar, no bank access): ~4x less LDS
// pressure and no barrier of any kind.  Lists are padded to 64*KPL with +inf keys, so every comparator is unconditional.
// value of lane (lane ^ LM): a DPP move on the VALU where the pattern exists inside a 16-lane row (no LDS round trip: the
// network is a chain of dependent exchanges, and a wave spent half its life waiting for ds_bpermute results), the LDS crossbar
// otherwise.  xor 1, 2, 3 = quad permutes; 7 / 15 = half-row / row mirror; 8 = rotate by 8; 4 = two shifted moves with
// complementary bank masks.
template <uint32_t LM>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    const int x = (int)v;
    // (old = 0 with bound_ctrl: every lane of these patterns has a source inside its row, so `old` is never read -- but naming x as `old` made the
    //  compiler copy x into the destination first: one v_mov_b32 + a DPP hazard s_nop per exchanged word, 897 + 905 of them in k_sort_multi)
    if (LM == 1u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    if (LM == 2u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    if (LM == 3u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    if (LM == 7u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    if (LM == 15u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); // row_mirror
    if (LM == 8u) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);  // row_ror:8
    if (LM == 4u) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0, 2: lane <- lane + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);  // row_shr:4 into banks 1, 3: lane <- lane - 4
    }
    return (uint32_t)__shfl_xor(x, (int)LM, 64);
}

// Order of two keys.  A key is (float bits of a depth > 0.2) << 32 | Gaussian id: read as an IEEE DOUBLE it is a positive, finite, normal number
// (sign 0; the double's exponent field = the float's exponent + its three top mantissa bits, never all ones for a finite or infinite float depth; NaN
// depths are culled by k_preprocess), and positive doubles order exactly like their bit patterns.  v_cmp_lt_f64 runs at the full vector rate on
// gfx950; v_cmp_lt_u64 does not -- with one 64-bit integer compare per compare-exchange the sort kernels were bound by it (round 4: k_sort_multi
// 500 us at 15,000 lists of ~1,700 keys, ~11 cycles per instruction on average).  The padding key is +infinity (0x7FF0...0): above every real
// key, equal to itself, never NaN.
constexpr uint64_t SORT_PAD = 0x7FF0000000000000ull;
__device__ __forceinline__ bool key_lt(uint64_t a, uint64_t b) { return __longlong_as_double((long long)a) < __longlong_as_double((long long)b); }
// min / max of two keys as ONE instruction each (v_min_f64 / v_max_f64: bit-exact on positive finite doubles and +infinity).  A compare-exchange
// of two keys held by the same lane is then 2 instructions instead of a compare + a hazard wait + 4 selects through VCC -- the sort kernels are
// instruction-issue bound (round 4 counters: 1.78e8 VALU instructions per k_sort_multi launch, the vector ALUs busy 2/3 of the kernel).  Inline
// asm because fmin() / fmax() come with a canonicalising v_max_f64 x, x, x per operand (IEEE mode quiets signalling NaNs; there are none here).
__device__ __forceinline__ uint64_t key_min(uint64_t a, uint64_t b) {
    double r;
    asm volatile("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (uint64_t)__double_as_longlong(r);
}
__device__ __forceinline__ uint64_t key_max(uint64_t a, uint64_t b) {
    double r;
    asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (uint64_t)__double_as_longlong(r);
}

// one stage of the ascending-only bitonic network (KB = block level, ST = stage inside it; everything a compile-time constant)
template <int KPL, int KB, int ST>
__device__ __forceinline__ void sort_stage(uint64_t (&key)[KPL], int lane) {
    constexpr int LOGK = KPL == 1 ? 0 : KPL == 2 ? 1 : KPL == 4 ? 2 : KPL == 8 ? 3 : 4;
    // ST = 0: flip (mask = 2^KB - 1); ST > 0: half-cleaner of stride 2^(KB-1-ST)
    constexpr uint32_t mask = ST == 0 ? ((1u << KB) - 1u) : (1u << (KB - 1 - ST));
    constexpr uint32_t top = ST == 0 ? (1u << (KB - 1)) : mask;  // the element without this bit is the lower of its pair
    constexpr uint32_t rmask = mask & (KPL - 1), lmask = mask >> LOGK;
    if constexpr (lmask == 0) {  // both elements live in this lane's registers
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int r2 = r ^ (int)rmask;
            if (r2 > r) {
                const uint64_t a = key[r], b = key[r2];
                key[r] = key_min(a, b);
                key[r2] = key_max(a, b);
            }
        }
    } else {  // partner element lives in lane ^ lmask, register r ^ rmask
        const bool lower = ((uint32_t)lane & (top >> LOGK)) == 0;
        uint64_t other[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint64_t mine = key[r ^ (int)rmask];
            const uint32_t lo = lane_xor<lmask>((uint32_t)mine);
            const uint32_t hi = lane_xor<lmask>((uint32_t)(mine >> 32));
            other[r] = ((uint64_t)hi << 32) | lo;
        }
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint64_t a = key[r], b = other[r];
            // lower element of the pair keeps the minimum, upper the maximum: both are computed (one instruction each) and the lane's role -- a
            // loop-invariant mask in SGPRs -- selects; no compare, so nothing is serialised through VCC
            const uint64_t mn = key_min(a, b), mx = key_max(a, b);
            key[r] = lower ? mn : mx;
        }
    }
}
template <int KPL, int LOGN, int KB, int ST>
__device__ __forceinline__ void sort_stages_from(uint64_t (&key)[KPL], int lane) {
    sort_stage<KPL, KB, ST>(key, lane);
    if constexpr (ST + 1 < KB) sort_stages_from<KPL, LOGN, KB, ST + 1>(key, lane);
    else if constexpr (KB < LOGN) sort_stages_from<KPL, LOGN, KB + 1, 0>(key, lane);
}

template <int KPL>
__device__ __forceinline__ void sort_wave_regs(const uint64_t *__restrict__ seg, uint32_t n, uint32_t *__restrict__ out, int lane) {
    constexpr int LOGK = KPL == 1 ? 0 : KPL == 2 ? 1 : KPL == 4 ? 2 : KPL == 8 ? 3 : 4;
    constexpr int LOGN = LOGK + 6;
    uint64_t key[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = (uint32_t)lane * KPL + r;
        key[r] = e < n ? seg[e] : SORT_PAD;
    }
    sort_stages_from<KPL, LOGN, 1, 0>(key, lane);
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = (uint32_t)lane * KPL + r;
        if (e < n) out[e] = (uint32_t)key[r];
    }
}

// lists of 1..1024 keys (the common case: a body bin holds ~450): ONE WAVE per bin, keys in registers.
__global__ __launch_bounds__(64) void k_sort_wave(const uint32_t *__restrict__ bin_offset, const uint32_t *__restrict__ wg_order,
                                                  uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                                  const GsrHeader *__restrict__ hdr) {
    if (hdr->overflow || blockIdx.x >= hdr->num_busy_wgs) return;
    const uint32_t bin = wg_order[blockIdx.x];  // busy bins first
    const uint32_t off = bin_offset[bin], n = bin_offset[bin + 1] - off;
    if (n == 0 || n > 1024u) return;
    const uint64_t *seg = keys + off;
    uint32_t *out = point_list + off;
    const int lane = threadIdx.x;
    if (n <= 64u) sort_wave_regs<1>(seg, n, out, lane);
    else if (n <= 128u) sort_wave_regs<2>(seg, n, out, lane);
    else if (n <= 256u) sort_wave_regs<4>(seg, n, out, lane);
    else if (n <= 512u) sort_wave_regs<8>(seg, n, out, lane);
    else sort_wave_regs<16>(seg, n, out, lane);
}

// ---- lists of 1,025 .. 8,192 keys: 1 / 2 / 4 waves per list, 32 keys per lane in registers -------------------------------------------
// What freshly initialised networks produce (BASELINE configs 3 / 4 with random weights: scales at their 0.01 m clamp, ~20,000 bins with
// lists of 1,000-3,000 entries per 2048^2 view).  Rounds 1-3 sorted such a list with ONE 1,024-thread workgroup in 128 KiB of LDS, one key
// per thread and a workgroup barrier after each of the 66-78 network stages: barrier-latency bound, one list per CU at a time, 1.9 ms per
// view = 2.4 % of the HBM roof (profiles/r03_config3_kernel_stats.md).  Here the network of k_sort_wave is kept -- element e = t * 32 + r
// (t = thread of the workgroup, r = register), partners e ^ mask -- so of the 66 stages of a 2,048-key sort 45 are register-to-register
// and 21 are lane exchanges (DPP / LDS crossbar), with no barrier and no LDS memory traffic at all for one wave; with 2 / 4 waves the 2 / 4
// stages whose mask reaches across waves go through a transposed LDS tile (conflict-free) behind two barriers.  The network is walked by
// RUNTIME loops over (level, stage) that dispatch into ~20 straight-line stage bodies (one per register / lane mask): ~35 KiB of code instead
// of ~120 KiB for the fully unrolled form, which would stream through the instruction cache once per list.
// Keys arrive unsorted, so they are loaded coalesced (register r = keys r * T + t) and only the sorted ids are written in element order.
constexpr int SM_KPL = 32, SM_LOGK = 5;

template <uint32_t LM>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v) {
    return ((uint64_t)lane_xor<LM>((uint32_t)(v >> 32)) << 32) | lane_xor<LM>((uint32_t)v);
}

// both elements in this lane's registers: r <-> r ^ RM (ascending: the lower register index keeps the minimum)
template <int RM>
__device__ __forceinline__ void sm_inreg(uint64_t (&key)[SM_KPL]) {
#pragma unroll
    for (int r = 0; r < SM_KPL; r++) {
        const int r2 = r ^ RM;
        if (r2 > r) {
            const uint64_t a = key[r], b = key[r2];
            key[r] = key_min(a, b);
            key[r2] = key_max(a, b);
        }
    }
}
// partner in lane ^ LM, register r ^ RM (RM = 0: half-cleaner, RM = 31: flip).  Pairs of registers are handled together so that only two
// exchanged keys are live at a time (the k_sort_wave form materialises all KPL of them first: 64 more VGPRs at 32 keys per lane).
template <uint32_t LM, int RM>
__device__ __forceinline__ void sm_cross(uint64_t (&key)[SM_KPL], bool lower) {
#pragma unroll
    for (int r = 0; r < SM_KPL; r++) {
        const int r2 = r ^ RM;
        if (r2 < r) continue;
        const uint64_t mine = key[r], mine2 = key[r2];
        const uint64_t got = lane_xor64<LM>(mine2);           // the partner lane's register r2 pairs with my register r
        const uint64_t mn = key_min(mine, got), mx = key_max(mine, got);  // both, unconditionally; the lane's role (a loop-invariant SGPR mask) selects
        key[r] = lower ? mn : mx;
        if (r2 != r) {
            const uint64_t got2 = lane_xor64<LM>(mine);
            const uint64_t mn2 = key_min(mine2, got2), mx2 = key_max(mine2, got2);
            key[r2] = lower ? mn2 : mx2;
        }
    }
}
__device__ __forceinline__ void sm_inreg_switch(uint32_t rm, uint64_t (&key)[SM_KPL]) {
    switch (rm) {
        case 1: sm_inreg<1>(key); break;
        case 2: sm_inreg<2>(key); break;
        case 3: sm_inreg<3>(key); break;
        case 4: sm_inreg<4>(key); break;
        case 7: sm_inreg<7>(key); break;
        case 8: sm_inreg<8>(key); break;
        case 15: sm_inreg<15>(key); break;
        case 16: sm_inreg<16>(key); break;
        default: sm_inreg<31>(key); break;
    }
}
__device__ __forceinline__ void sm_clean_switch(int jl /* lane bit */, uint64_t (&key)[SM_KPL], int lane) {
    const bool lower = ((lane >> jl) & 1) == 0;
    switch (jl) {
        case 0: sm_cross<1u, 0>(key, lower); break;
        case 1: sm_cross<2u, 0>(key, lower); break;
        case 2: sm_cross<4u, 0>(key, lower); break;
        case 3: sm_cross<8u, 0>(key, lower); break;
        case 4: sm_cross<16u, 0>(key, lower); break;
        default: sm_cross<32u, 0>(key, lower); break;
    }
}
__device__ __forceinline__ void sm_flip_switch(int kl /* lane bits the flip mirrors: 1..6 */, uint64_t (&key)[SM_KPL], int lane) {
    const bool lower = ((lane >> (kl - 1)) & 1) == 0;
    switch (kl) {
        case 1: sm_cross<1u, 31>(key, lower); break;
        case 2: sm_cross<3u, 31>(key, lower); break;
        case 3: sm_cross<7u, 31>(key, lower); break;
        case 4: sm_cross<15u, 31>(key, lower); break;
        case 5: sm_cross<31u, 31>(key, lower); break;
        default: sm_cross<63u, 31>(key, lower); break;
    }
}
// a stage whose mask reaches across the waves of the workgroup: through LDS, register-major (xch[r * T + t]: lanes of a wave are adjacent words,
// and t ^ tmask permutes whole 64-groups or lanes inside one -- no bank conflicts either way)
template <int NW>
__device__ __forceinline__ void sm_lds_stage(uint64_t (&key)[SM_KPL], uint64_t *xch, int t, uint32_t mask, uint32_t top) {
    constexpr int T = NW * 64;
    const uint32_t rm = mask & (SM_KPL - 1), tm = mask >> SM_LOGK;
    const bool lower = (((uint32_t)t << SM_LOGK) & top) == 0;
    __syncthreads();  // the previous exchange's reads are done
#pragma unroll
    for (int r = 0; r < SM_KPL; r++) xch[r * T + t] = key[r];
    __syncthreads();
    const uint32_t tp = (uint32_t)t ^ tm;
    if (rm == 0) {
#pragma unroll
        for (int r = 0; r < SM_KPL; r++) {
            const uint64_t got = xch[r * T + tp], mine = key[r];
            const uint64_t mn = key_min(mine, got), mx = key_max(mine, got);
            key[r] = lower ? mn : mx;
        }
    } else {  // flip: rm == 31
#pragma unroll
        for (int r = 0; r < SM_KPL; r++) {
            const uint64_t got = xch[(r ^ (SM_KPL - 1)) * T + tp], mine = key[r];
            const uint64_t mn = key_min(mine, got), mx = key_max(mine, got);
            key[r] = lower ? mn : mx;
        }
    }
}

template <int NW>
__device__ __forceinline__ void sort_multi(const uint64_t *__restrict__ seg, uint32_t n, uint32_t *__restrict__ out, uint64_t *xch, int t) {
    constexpr int T = NW * 64;
    constexpr int LOGN = SM_LOGK + 6 + (NW == 1 ? 0 : NW == 2 ? 1 : 2);
    const int lane = t & 63;
    uint64_t key[SM_KPL];
#pragma unroll
    for (int r = 0; r < SM_KPL; r++) {
        const uint32_t e = (uint32_t)(r * T + t);  // any assignment of the unsorted keys to elements will do: coalesced
        key[r] = e < n ? seg[e] : SORT_PAD;
    }
    for (int kb = 1; kb <= LOGN; kb++) {  // merge sorted runs of 2^(kb-1) into runs of 2^kb: one flip, then half-cleaners of stride 2^j
        if (kb <= SM_LOGK) sm_inreg_switch((1u << kb) - 1u, key);
        else if (kb <= SM_LOGK + 6) sm_flip_switch(kb - SM_LOGK, key, lane);
        else if (NW > 1) sm_lds_stage<NW>(key, xch, t, (1u << kb) - 1u, 1u << (kb - 1));
        for (int j = kb - 2; j >= 0; j--) {
            if (j < SM_LOGK) sm_inreg_switch(1u << j, key);
            else if (j < SM_LOGK + 6) sm_clean_switch(j - SM_LOGK, key, lane);
            else if (NW > 1) sm_lds_stage<NW>(key, xch, t, 1u << j, 1u << j);
        }
    }
#pragma unroll
    for (int r = 0; r < SM_KPL; r++) {
        const uint32_t e = (uint32_t)t * SM_KPL + r;
        if (e < n) out[e] = (uint32_t)key[r];
    }
}

// A persistent grid strides over the BUSY bins in work order (wg_order: the longest lists first, so the stride deals them out evenly) and
// sorts those whose length falls into this launch's class: (1024 << (NW/2)) < n <= 2048 * NW ... i.e. NW = 1: 1,025-2,048, 2: -4,096, 4: -8,192.
#ifndef GSR_SORT_MULTI_WAVES
#define GSR_SORT_MULTI_WAVES 3  // waves per SIMD asked of the compiler for the one- and two-wave classes (168 VGPRs)
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 4 ? 2 : GSR_SORT_MULTI_WAVES))) void k_sort_multi(const uint32_t *__restrict__ bin_offset, const uint32_t *__restrict__ wg_order,
                                                        const uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                                        const GsrHeader *__restrict__ hdr) {
    __shared__ uint64_t xch[NW > 1 ? NW * 64 * SM_KPL : 1];
    constexpr uint32_t HI = 2048u * NW, LO = NW == 1 ? 1024u : HI / 2u;
    if (hdr->overflow || hdr->max_tile_count <= LO) return;
    const uint32_t busy = hdr->num_busy_wgs;
    for (uint32_t pos = blockIdx.x; pos < busy; pos += gridDim.x) {
        const uint32_t bin = wg_order[pos];
        const uint32_t off = bin_offset[bin], n = bin_offset[bin + 1] - off;  // workgroup-uniform
        if (n > LO && n <= HI) sort_multi<NW>(keys + off, n, point_list + off, xch, threadIdx.x);
    }
}

// lists longer than 8,192 keys: a persistent grid of 1024-thread workgroups (128 KiB LDS each: one resident per CU) strides over the bins
// and picks them up (launching one big workgroup per bin just to exit cost ~18 us at 16,384 bins).  Rare in the synthetic configs (the
// launch is left out until a long list has been seen on the device), the NORM with freshly initialised networks: BASELINE config 3 with random
// weights has ~20,000 bins of 1,000-3,000 entries per view, and the 64-workgroup grid of rounds 1-2 took 6.0 ms of a 7.6 ms render there
// (profiles/r03_config3_kernel_stats.md) -- hence up to 1,024 workgroups: every CU busy, and the stride spreads long and short lists
__global__ __launch_bounds__(1024) void k_sort_large(int NB, const uint32_t *__restrict__ bin_offset, uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ point_list, const GsrHeader *__restrict__ hdr) {
    __shared__ uint64_t sk[16384];
    if (hdr->overflow || hdr->max_tile_count <= 8192u) return;
    for (int b = blockIdx.x; b < NB; b += gridDim.x) {
        const uint32_t off = bin_offset[b], n = bin_offset[b + 1] - off;  // wave-uniform
        if (n > 8192u) {
            sort_one_bin<1024, 16384, true>(sk, off, n, keys, point_list, threadIdx.x);
            __syncthreads();
        }
    }
}

}  // namespace

void gsr_launch_scan(const uint32_t *bin_count, uint32_t *bin_offset, uint32_t *bin_cursor, uint32_t *wg_order, uint4 *scan_part, int NB, int bx, int by,
                     int64_t cap, GsrHeader *hdr, uint32_t *gpart, int n_gblocks, uint32_t *host_hdr, uint32_t host_seq, bool no_large_sort,
                     uint32_t order_hint, hipStream_t s) {
    const int NT = ((bx + 7) / 8) * ((by + 7) / 8) * 64;  // indices of the patch order (>= NB: ragged patches at the grid edge)
    const int nblocks = ((NT > NB ? NT : NB) + SB - 1) / SB;
    // (Tried: ONE 1,024-thread workgroup owning 16 indices per thread for grids of up to 16,384 bins -- no partials, no flags.  Bit-identical
    //  outputs, but 40.8 us against the fused form's 14.6: sixteen dependent rounds of patch-order index arithmetic and gathers on one CU cost far
    //  more than the release / acquire round trip they avoid.  Removed; profiles/r03_issue_probes.md section 4.)
    if (nblocks <= GSR_SCAN_FUSE_MAX) {
        hipLaunchKernelGGL(k_scan_b<true>, dim3(nblocks), dim3(SB), 0, s, bin_count, scan_part, bin_offset, bin_cursor, wg_order, NB, bx, by, nblocks, cap,
                           hdr, gpart, n_gblocks, host_hdr, host_seq, no_large_sort ? 1 : 0, order_hint);
        return;
    }
    hipLaunchKernelGGL(k_scan_a, dim3(nblocks), dim3(SB), 0, s, bin_count, scan_part, NB, bx, by, order_hint);
    hipLaunchKernelGGL(k_scan_b<false>, dim3(nblocks), dim3(SB), 0, s, bin_count, scan_part, bin_offset, bin_cursor, wg_order, NB, bx, by, nblocks, cap, hdr,
                       gpart, n_gblocks, host_hdr, host_seq, no_large_sort ? 1 : 0, order_hint);
}

void gsr_launch_scatter(int P, const uint32_t *row_range, int bx, const GsrSplat *splats, const uint32_t *hitmask, const uint32_t *wg_tab, uint32_t *bin_cursor, uint64_t *keys, const GsrHeader *hdr,
                        const uint32_t *goff, const uint32_t *gpart, uint8_t *inst_valid, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3((P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS), dim3(GSR_BIN_THREADS), 0, s, P, row_range, bx, splats, hitmask, wg_tab, bin_cursor, keys, hdr,
                       goff, gpart, inst_valid);
}

void gsr_launch_sort(int NB, const uint32_t *bin_offset, const uint32_t *wg_order, uint64_t *keys, uint32_t *point_list,
                     const GsrHeader *hdr, bool no_large_sort, hipStream_t s) {
    if (NB <= 0) return;
    hipLaunchKernelGGL(k_sort_wave, dim3(NB), dim3(64), 0, s, bin_offset, wg_order, keys, point_list, hdr);
    if (no_large_sort) return;  // the scan has turned any list longer than 1024 into an overflow (nothing downstream runs)
    // 1,025 .. 8,192 keys: 1 / 2 / 4 waves per list, keys in registers (each launch returns at once when the view's longest list is below its class)
    // Grid of the one-wave class: MANY more workgroups than the chip holds at once, so that the hardware dispatcher deals the lists out dynamically
    // (measured at R = 3e7, ~15,000 lists of ~1,700 keys: 1,091 / 651 / 571 / 470 us for 1,024 / 2,048 / 4,096 / 6,144 workgroups striding over the busy
    // bins -- a workgroup that drew five long lists is what the others wait for).  A workgroup without a list in its class costs a few scalar loads.
    // GPSGS_DEBUG_SORT_GRID=<n> overrides it (development).
    static int g1 = -1;
    if (g1 < 0) { const char *e = getenv("GPSGS_DEBUG_SORT_GRID"); g1 = e ? atoi(e) : 16384; if (g1 <= 0) g1 = 16384; }
    const int busy_max = NB;
    // (the same holds for the two- and four-wave classes: with the real networks' Gaussians -- lists up to ~3,000 keys -- the two-wave class is the
    //  largest: 409 us on 1,536 workgroups, profiles/r04_config3_kernel_stats.md)
    hipLaunchKernelGGL(k_sort_multi<4>, dim3(busy_max < 2048 ? busy_max : 2048), dim3(256), 0, s, bin_offset, wg_order, keys, point_list, hdr);
    hipLaunchKernelGGL(k_sort_multi<2>, dim3(busy_max < 8192 ? busy_max : 8192), dim3(128), 0, s, bin_offset, wg_order, keys, point_list, hdr);
    hipLaunchKernelGGL(k_sort_multi<1>, dim3(busy_max < g1 ? busy_max : g1), dim3(64), 0, s, bin_offset, wg_order, keys, point_list, hdr);
    hipLaunchKernelGGL(k_sort_large, dim3(NB < 1024 ? NB : 1024), dim3(1024), 0, s, NB, bin_offset, keys, point_list, hdr);
}
