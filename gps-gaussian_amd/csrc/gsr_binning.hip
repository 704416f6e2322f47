// gsr_binning.hip -- bin assignment and per-bin depth ordering for gfx950.
//
// Upstream builds one global list of (tile<<32 | depth) keys and radix-sorts all R of them through HBM (~6 passes
// of 24 B/instance) after a blocking D2H read of R.  Here the bin part of the key is
// resolved by construction instead of by sorting:
//   k_scan_a/b two-phase parallel exclusive scan of the per-bin counts written by k_preprocess (normally fused into ONE launch: the blocks exchange
//              their partials as self-validating 64-bit atomics -- no release / acquire fence: round 5) -> bin_offset, cursors, R, overflow flag;
//              R is published to the host from here.  The same pass emits wg_order: compositing workgroups with work first (so every CU starts
//              on real work and the empty ones drain in the gaps), longest lists first, patch order inside a class.
//   k_scatter  each Gaussian drops (depth_bits<<32 | id) into its bins' segments: slot = bin_offset[bin] + the base k_preprocess' count atomic
//              returned for this workgroup (recorded in wg_tab) + the rank inside the workgroup (LDS atomic).  No global atomic, nothing
//              recounted or re-tested (per-cell hit masks from k_preprocess).  512 threads x 2 Gaussians: every workgroup of a 600 k-Gaussian
//              view is resident at once (one generation instead of two: the kernel is a latency chain, not a throughput problem).
//   k_sort_*   one WAVE per bin sorts its segment with an ascending-only bitonic network on the 64-bit key, keys in
//              registers, exchanges by DPP / LDS crossbar (lists <= 1024); longer lists get a 1024-thread workgroup in LDS.  Keys are unique (id in the low word) so the result is deterministic and equals
//              upstream's stable radix order: depth ascending, ties by Gaussian index.
//              Segments longer than the LDS capacity fall back to the same network run in global memory.
// HBM traffic: 8 B written + 8 B read + 4 B written per instance, versus >= 144 B for the global radix sort.
#include "gsr_common.h"
#include "gsr_sort_wave.h"

namespace {

// Work order of the compositing / sorting waves: the bin grid enumerated in PATCHES of 8x8 bins (64x64 pixels; row-major inside a
// patch, patches row-major).  A Gaussian is listed in ~3 neighbouring bins, mostly a 2x2 block: in this order the block's bins are
// (with probability ~0.77) within 64 consecutive entries, which xcd_list_pos() hands to ONE XCD -- one L2 then fetches the splat
// record once for all of them and merges the gradient records the compositing backward scatters into the Gaussian's (contiguous)
// slots.  Image order (runs of 64x1 bins) shared only the horizontal neighbours.  -1: the index has no bin (ragged grid edge).
__device__ __forceinline__ int gsr_tiled_bin(uint32_t t, int bx, int by) {
    const uint32_t pgx = ((uint32_t)bx + 7u) >> 3;
    const uint32_t p = t >> 6, w = t & 63u;
    const uint32_t x = (p % pgx) * 8u + (w & 7u), y = (p / pgx) * 8u + (w >> 3);
    return (x < (uint32_t)bx && y < (uint32_t)by) ? (int)(y * (uint32_t)bx + x) : -1;
}
// Work classes of a bin, by the length of its list relative to the longest list an earlier, similar view produced (GsrViewExt.order_hint; 0 =
// unknown = every busy bin in class 0 = plain patch order): 0 = more than 1/2 of it, 1 = more than 1/4, 2 = any other busy bin, 3 = idle.
// Dispatched in that order: a bin is ONE wave's sequential job and a SIMD gets only ~5 of them per kernel, so a 900-entry list that starts late
// is what the other SIMDs end up waiting for (longest-processing-time-first, coarsely).  A performance hint only: every bin is handled exactly
// once whatever its position.
__device__ __forceinline__ int gsr_work_class(int wb, uint32_t wc, uint32_t hint) {
    return wb < 0 ? -1 : (wc > (hint >> 1) ? 0 : (wc > (hint >> 2) ? 1 : (wc > 0u ? 2 : 3)));
}

constexpr int SB = GSR_SCAN_BLOCK;  // 1024 bins per scan block

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

__device__ __forceinline__ uint32_t class_w0(int cls) { return cls == 0 ? 1u : (cls == 1 ? 0x10000u : 0u); }
__device__ __forceinline__ uint32_t class_w1(int cls) { return cls == 2 ? 1u : (cls == 3 ? 0x10000u : 0u); }

// phase A: per block of 1024 indices -> part[2 blk] = {sum of counts (bins in image order), w0 of the block's indices in WORK order
// (tiled_bin), max count, -}, part[2 blk + 1].x = their w1
__global__ __launch_bounds__(SB) void k_scan_a(const uint32_t *__restrict__ bin_count, const uint32_t *__restrict__ bin_count_fb, uint4 *__restrict__ part, int NB,
                                               int bx, int by, uint32_t hint) {
    __shared__ uint32_t red[4][SB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x * SB + tid;
    const uint32_t c = b < NB ? bin_count[(size_t)b * GSR_CPAD] + bin_count_fb[(size_t)b * GSR_CPAD] : 0u;
    const int wb = gsr_tiled_bin((uint32_t)b, bx, by);
    const uint32_t wc = wb >= 0 ? bin_count[(size_t)wb * GSR_CPAD] + bin_count_fb[(size_t)wb * GSR_CPAD] : 0u;
    const int cls = gsr_work_class(wb, wc, hint);
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
__global__ __launch_bounds__(SB) void k_scan_b(const uint32_t *__restrict__ bin_count, const uint32_t *__restrict__ bin_count_fb, uint4 *__restrict__ part,
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
    const uint32_t c_rec = b < NB ? bin_count[(size_t)b * GSR_CPAD] : 0u;  // instances with a recorded base (k_preprocess' returning atomics)
    const uint32_t c = c_rec + (b < NB ? bin_count_fb[(size_t)b * GSR_CPAD] : 0u);  // + those of workgroups whose bins did not fit the table
    const int wb = gsr_tiled_bin((uint32_t)b, bx, by);  // the bin this thread places in the work order
    const uint32_t wc = wb >= 0 ? bin_count[(size_t)wb * GSR_CPAD] + bin_count_fb[(size_t)wb * GSR_CPAD] : 0u;
    const int cls = gsr_work_class(wb, wc, hint);
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
        // Each partial travels as TWO self-validating 64-bit words {sum | class 0 | class 1 | flag} and {max | class 2 | idle | flag}, written and
        // read with RELAXED agent-scope atomics: data and ready flag share a word, so no ordering between addresses is needed -- and no release /
        // acquire fence.  (A device-scope release is a write-back of the XCD's L2, which at this point holds k_preprocess' freshly written splat
        // records, an acquire invalidates it: rounds 1-4 paid for both in the middle of a 16-workgroup latency chain -- measured 14.8 -> 10.8 us
        // for this kernel.)  Class counts <= 1,024: 11 bits.
        if (tid == 0) {
            uint32_t ts = 0, tb = 0, tm = 0, ti = 0;
            for (int w = 0; w < SB / 64; w++) { ts += red[0][w]; tb += red[1][w]; tm = red[2][w] > tm ? red[2][w] : tm; ti += red[3][w]; }
            unsigned long long *me = reinterpret_cast<unsigned long long *>(part + 2 * blockIdx.x);
            const unsigned long long A = (unsigned long long)ts | ((unsigned long long)(tb & 0xffffu) << 32) | ((unsigned long long)(tb >> 16) << 43) | (1ull << 63);
            const unsigned long long B = (unsigned long long)tm | ((unsigned long long)(ti & 0xffffu) << 32) | ((unsigned long long)(ti >> 16) << 43) | (1ull << 63);
            __hip_atomic_store(me, A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(me + 1, B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        slot_scan();  // after this block's partial is out, before the others' are needed
        if (tid < nblocks) {  // wait for every block's partial (all blocks are resident: nblocks <= GSR_SCAN_FUSE_MAX)
            unsigned long long *other = reinterpret_cast<unsigned long long *>(part + 2 * tid);
            unsigned long long A, B;
            while (((A = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) == 0ull) __builtin_amdgcn_s_sleep(1);
            while (((B = __hip_atomic_load(other + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) == 0ull) __builtin_amdgcn_s_sleep(1);
            sp[tid] = make_uint4((uint32_t)A, (uint32_t)((A >> 32) & 0x7ffu) | ((uint32_t)((A >> 43) & 0x7ffu) << 16), (uint32_t)B, 1u);
            sp_idle[tid] = (uint32_t)((B >> 32) & 0x7ffu) | ((uint32_t)((B >> 43) & 0x7ffu) << 16);
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
            // System-scope write-through stores, waited for (vmcnt counts stores on gfx9: zero = acknowledged by the fabric), THEN the sequence word the
            // host polls -- not a system-scope release fence, which would write this XCD's whole L2 back first.
            const uint32_t hv[7] = {(uint32_t)tot_sum, (uint32_t)(tot_sum >> 32), ovf_b ? 1u : 0u, tot_max, tot_busy, tot_slots, hdr->num_points};
#pragma unroll
            for (int k = 0; k < 7; k++) __hip_atomic_store(host_hdr + k, hv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(host_hdr + 7, host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ONE block scan of three words (this block's bin counts and the two packed class counters): the kernel is a latency chain, and three scans
    // were six barriers
    uint32_t off, r0, r1;
    {
        __shared__ uint32_t w3[3][SB / 64];
        const int lane = tid & 63, wid = tid >> 6;
        const uint32_t v[3] = {c, class_w0(cls), class_w1(cls)};
        uint32_t x[3] = {v[0], v[1], v[2]};
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t y = __shfl_up(x[k], d, 64);
                if (lane >= d) x[k] += y;
            }
        }
        __syncthreads();
        if (lane == 63) { w3[0][wid] = x[0]; w3[1][wid] = x[1]; w3[2][wid] = x[2]; }
        __syncthreads();
        uint32_t wo[3] = {0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < SB / 64; w++)
#pragma unroll
            for (int k = 0; k < 3; k++) wo[k] += w < wid ? w3[k][w] : 0u;
        off = pre_sum + wo[0] + x[0] - v[0];
        r0 = wo[1] + x[1] - v[1];
        r1 = wo[2] + x[2] - v[2];
    }
    if (b < NB) {
        bin_offset[b] = off;
        bin_cursor[(size_t)b * GSR_CPAD] = off + c_rec;  // where the instances of incoherent workgroups start: behind the recorded ones
    }
    // work-ordered workgroup list: the classes in order (longest lists first, idle bins last), patch order (tiled_bin) inside each class
    if (wb >= 0) {
        const uint32_t pos = cls == 0 ? pre[0] + (r0 & 0xffffu)
                           : cls == 1 ? tot[0] + pre[1] + (r0 >> 16)
                           : cls == 2 ? tot[0] + tot[1] + pre[2] + (r1 & 0xffffu)
                                      : tot_busy + pre[3] + (r1 >> 16);
        wg_order[pos] = (uint32_t)wb;
    }
}


constexpr int SC_PER = 2;                                // Gaussians per thread: a workgroup owns the Gaussians of ONE k_preprocess workgroup (its recorded table)
constexpr int SC_THREADS = GSR_BIN_THREADS / SC_PER;     // threads of a scatter workgroup

// ---- DIRECT lists (GsrBins::cap > 0; round 6) ----------------------------------------------------------------------------------------------------------
// With a fixed-capacity segment per bin nothing downstream needs the exclusive scan of the counts any more, and the scatter needs nothing but the
// counters k_preprocess left.  What is left of k_scan_b rides in the SCATTER launch, in workgroups in front of the scattering ones:
//   - the totals (R, longest list, gradient slots, overflow) and the header, also towards the host: scatter_totals_block;
//   - the work order of the compositing waves and the slot prefix of the gradient records: scan_wave_direct, one wave per 64 indices of the patch
//     order.  The waves exchange their partials as one self-validating 64-bit word each, like k_scan_b's blocks (no fences); the scattering
//     workgroups never wait for anything, so the polling scan waves cannot deadlock whatever the dispatch order (<= 1,024 of them).
// The forward chain is then k_preprocess (count) -> k_scatter<true> -> k_sort_wave -> compositing: the ~10 us latency chain of the scan runs beside the
// scatter instead of in front of it, and the scatter gathers no offsets.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ unsigned long long poll_word(unsigned long long *w) {
    unsigned long long v;
    while (((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) == 0ull) __builtin_amdgcn_s_sleep(1);
    return v;
}


__device__ __forceinline__ void scan_wave_direct(int blk, int nscan, int lane, int bx, int by, const uint32_t *__restrict__ bin_count, const uint32_t *__restrict__ bin_count_fb,
                                                 uint32_t *__restrict__ wg_order, unsigned long long *__restrict__ part /* 1 word per scan wave */, GsrHeader *__restrict__ hdr,
                                                 const uint32_t *__restrict__ gpart, uint32_t *__restrict__ gprefix, int n_gblocks, uint32_t hint) {
    const int b = blk * 64 + lane;
    const int wb = gsr_tiled_bin((uint32_t)b, bx, by);
    const uint32_t wc = wb >= 0 ? bin_count[wb] + bin_count_fb[wb] : 0u;
    const int cls = gsr_work_class(wb, wc, hint);
    const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
    // the slot prefix of the gradient records (training): the exclusive prefix (gprefix) of the per-workgroup slot counts k_preprocess left (gpart).  Scan wave
    // `blk` owns workgroups [blk * gper, (blk + 1) * gper) (gper a multiple of 64; 64 up to 4 M Gaussians at 1024^2) and publishes their sum with its
    // class counts.  
    const int gper = gpart ? (((n_gblocks + nscan - 1) / nscan + 63) & ~63) : 0;
    const int g0 = blk * gper, g1 = min(n_gblocks, g0 + gper);
    uint32_t gv0 = 0u, gsum = 0u;
    if (g0 < g1) {
        gv0 = g0 + lane < g1 ? gpart[g0 + lane] : 0u;
        gsum = gv0;
        for (int k = g0 + 64 + lane; k < g1; k += 64) gsum += gpart[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) gsum += (uint32_t)__shfl_xor((int)gsum, d, 64);
    }
    if (lane == 0) {  // class counts <= 64: 7 bits each; the slot sum: 32 bits; + the ready bit: one self-validating word
        const unsigned long long A = (unsigned long long)__popcll(m0) | ((unsigned long long)__popcll(m1) << 7) | ((unsigned long long)__popcll(m2) << 14) |
                                     ((unsigned long long)__popcll(m3) << 21) | ((unsigned long long)gsum << 28) | (1ull << 63);
        __hip_atomic_store(part + blk, A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every scan wave needs the class totals of all of them and the prefix of those in front of it
    uint32_t t[4] = {0u, 0u, 0u, 0u}, pre[4] = {0u, 0u, 0u, 0u}, gpre = 0u;
    for (int i0 = 0; i0 < nscan; i0 += 256) {  // four words per lane requested together (one round trip for the 256 waves of a 1024^2 view), then polled
        unsigned long long A[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u + lane;
            A[u] = i < nscan ? __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull << 63;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u + lane;
            if ((A[u] >> 63) == 0ull) A[u] = poll_word(part + i);
            const uint32_t n0 = (uint32_t)A[u] & 0x7fu, n1 = (uint32_t)(A[u] >> 7) & 0x7fu, n2 = (uint32_t)(A[u] >> 14) & 0x7fu, n3 = (uint32_t)(A[u] >> 21) & 0x7fu;
            t[0] += n0; t[1] += n1; t[2] += n2; t[3] += n3;
            if (i < blk) { pre[0] += n0; pre[1] += n1; pre[2] += n2; pre[3] += n3; gpre += (uint32_t)(A[u] >> 28); }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t[k] += (uint32_t)__shfl_xor((int)t[k], d, 64);
            pre[k] += (uint32_t)__shfl_xor((int)pre[k], d, 64);
        }
        gpre += (uint32_t)__shfl_xor((int)gpre, d, 64);
    }
    if (g0 < g1) {
        uint32_t carry = gpre;
        for (int base = g0; base < g1; base += 64) {
            const int k = base + lane;
            const uint32_t v = base == g0 ? gv0 : (k < g1 ? gpart[k] : 0u);
            const uint32_t x = wave_incl_scan(v, lane);
            if (k < g1) gprefix[k] = carry + x - v;
            carry += (uint32_t)__shfl((int)x, 63, 64);
        }
    }
    const uint32_t tot_busy = t[0] + t[1] + t[2];
    if (blk == 0 && lane == 0) hdr->num_busy_wgs = tot_busy;
    // work order: the classes in order (longest lists first, idle bins last), patch order inside a class
    if (wb >= 0) {
        const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
        const uint32_t r = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        const uint32_t pos = cls == 0 ? pre[0] + r : cls == 1 ? t[0] + pre[1] + r : cls == 2 ? t[0] + t[1] + pre[2] + r : tot_busy + pre[3] + r;
        wg_order[pos] = (uint32_t)wb;
    }
}

// DIRECT lists (GsrBins::cap > 0): the first SC_TOTALS workgroups of the launch do not scatter -- they add up the view from the counters k_preprocess
// left (R = sum of count + count_fb, the longest list, the gradient slots from the per-workgroup slot totals), merge their partial sums with three
// atomics each, and the last of them to finish publishes the header, also to the host: the early capacity notification leaves a few us into this
// launch, before a single key is sorted.  (Rejected on the way, docs/HISTORY.md round 6: the scatter pass INSIDE k_preprocess -- 17 us there against 12.6 us as
// its own kernel, the fused kernel is latency-bound -- and the totals from k_preprocess' last workgroup to arrive: +9 to +17 us of tail.)
constexpr uint32_t SC_TOTALS = 32;  // (512 bins each at 1024^2: two loads in flight per thread, then 4 atomics per block)
struct ScDirect {
    const uint32_t *bin_count, *bin_count_fb, *gpart;
    uint32_t *gprefix, *wg_order;
    unsigned long long *part;
    GsrHeader *hdr_w;
    uint32_t *host_hdr;
    int64_t slot_cap;
    uint32_t bin_cap, host_seq, hint;
    int NB, n_gblocks, by, nscan_wgs /* scan waves / 4 */;
};

__device__ __forceinline__ void scatter_totals_block(const ScDirect &d, uint32_t blk) {
    __shared__ unsigned long long t_sum[SC_THREADS / 64];
    __shared__ uint32_t t_max[SC_THREADS / 64], t_slots[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (d.NB + (int)SC_TOTALS - 1) / (int)SC_TOTALS, b0 = (int)blk * per, b1 = min(d.NB, b0 + per);
    unsigned long long sum = 0ull;
    uint32_t mx = 0u, slots = 0u;
#pragma unroll 2
    for (int b = b0 + tid; b < b1; b += SC_THREADS) {
        const uint32_t c = d.bin_count[(size_t)b * GSR_CPAD] + d.bin_count_fb[(size_t)b * GSR_CPAD];
        sum += c;
        mx = c > mx ? c : mx;
    }
    if (d.gpart) {
        const int gper = (d.n_gblocks + (int)SC_TOTALS - 1) / (int)SC_TOTALS, g0 = (int)blk * gper, g1 = min(d.n_gblocks, g0 + gper);
        for (int k = g0 + tid; k < g1; k += SC_THREADS) slots += d.gpart[k];
    }
#pragma unroll
    for (int s_ = 32; s_ >= 1; s_ >>= 1) {
        sum += (unsigned long long)__shfl_xor((long long)sum, s_, 64);
        slots += (uint32_t)__shfl_xor((int)slots, s_, 64);
        const uint32_t y = (uint32_t)__shfl_xor((int)mx, s_, 64);
        mx = y > mx ? y : mx;
    }
    if (lane == 0) { t_sum[wid] = sum; t_max[wid] = mx; t_slots[wid] = slots; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < SC_THREADS / 64; w++) { sum += t_sum[w]; slots += t_slots[w]; mx = t_max[w] > mx ? t_max[w] : mx; }
    GsrHeader *h = d.hdr_w;
    unsigned long long *acc_sum = reinterpret_cast<unsigned long long *>(&h->reserved[0]);  // (the header is zeroed with the counters; reserved[] is 8-byte aligned)
    if (sum) atomicAdd(acc_sum, sum);
    if (mx) atomicMax(&h->reserved[2], mx);
    if (slots) atomicAdd(&h->reserved[3], slots);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... this block's shares are in before it takes its ticket
    if (atomicAdd(&h->reserved[4], 1u) != SC_TOTALS - 1u) return;
    // the last totals block: every share is in (each block waited for its atomics before its ticket); read them past the caches
    const unsigned long long tsum = __hip_atomic_load(acc_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tmax = __hip_atomic_load(&h->reserved[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tot_slots = __hip_atomic_load(&h->reserved[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a list longer than the bins' capacity is reported like an overflow (max_tile_count > capacity tells the caller which): nothing is composited
    const bool ovf_b = tmax > d.bin_cap || (int64_t)tot_slots > d.slot_cap || h->row_overflow != 0u;
    h->num_rendered = tsum;
    h->overflow = ovf_b ? 1u : 0u;
    h->max_tile_count = tmax;
    h->num_slots = tot_slots;
    if (d.host_hdr) {  // as k_scan_b: write-through stores, acknowledged, then the sequence word the host polls
        const uint32_t hv[7] = {(uint32_t)tsum, (uint32_t)(tsum >> 32), ovf_b ? 1u : 0u, tmax, 0u /* busy bins: the scan waves of this launch are still counting them */, tot_slots, h->num_points};
#pragma unroll
        for (int k = 0; k < 7; k++) __hip_atomic_store(d.host_hdr + k, hv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(d.host_hdr + 7, d.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// (6 waves per SIMD: every workgroup of a 600 k-Gaussian view is resident at once)
template <bool DIRECT>
__global__ __launch_bounds__(SC_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_scatter(int P, const uint32_t *__restrict__ row_range, int bx, const GsrSplat *__restrict__ splats, const uint4 *__restrict__ binrec,
                                                       const uint32_t *__restrict__ wg_tab, const uint32_t *__restrict__ bin_offset, uint32_t *__restrict__ bin_cursor,
                                                       uint64_t *__restrict__ keys, const GsrHeader *__restrict__ hdr, ScDirect dd) {
    __shared__ uint32_t s_cnt[GSR_BLOCK_TAB], s_base[GSR_BLOCK_TAB];
    __shared__ int s_box[4];
    if (DIRECT && blockIdx.x < SC_TOTALS) {  // (workgroup-uniform)
        scatter_totals_block(dd, blockIdx.x);
        return;
    }
    if (DIRECT && blockIdx.x < SC_TOTALS + (uint32_t)dd.nscan_wgs) {
        scan_wave_direct((int)(blockIdx.x - SC_TOTALS) * (SC_THREADS / 64) + (int)(threadIdx.x >> 6), dd.nscan_wgs * (SC_THREADS / 64), (int)(threadIdx.x & 63), bx, dd.by, dd.bin_count,
                         dd.bin_count_fb, dd.wg_order, dd.part, dd.hdr_w, dd.gpart, dd.gprefix, dd.n_gblocks, dd.hint);
        return;
    }
    const uint32_t wg = DIRECT ? blockIdx.x - SC_TOTALS - (uint32_t)dd.nscan_wgs : blockIdx.x;  // the k_preprocess workgroup whose Gaussians this one scatters
    // The kernel is a chain of memory round trips (round-5 counters: two thirds of its wave cycles are spent waiting): every load that depends on
    // nothing but the launch is issued FIRST -- the Gaussians' records, the table header AND the table entries (unconditionally: entries beyond the
    // box are never used), the slot run -- and only then examined.  What is left behind them is one gather of bin_offset.
    const int i0 = (int)wg * GSR_BIN_THREADS;
    const uint32_t *tab = wg_tab + (size_t)wg * GSR_WG_TAB_WORDS;
    uint4 rec[SC_PER];
#pragma unroll
    for (int u = 0; u < SC_PER; u++) {
        const int i = i0 + u * SC_THREADS + (int)threadIdx.x;
        rec[u] = i < P ? binrec[i] : make_uint4(0u, 0u, 0u, 0u);  // (P = the launch capacity: rows behind a row-range view's last Gaussian hold neutral records)
    }
    constexpr int TPT_ALL = GSR_BLOCK_TAB / SC_THREADS;  // table entries per thread
    constexpr int TPT = TPT_ALL < 4 ? TPT_ALL : 4;       // ... of which the first 4 x SC_THREADS (every box a coherent view produces) are requested up front
    uint32_t tent[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) tent[k] = tab[4 + k * SC_THREADS + (int)threadIdx.x];
    const int tab_bx0 = (int)tab[0], tab_by0 = (int)tab[1], tab_bw = (int)tab[2], tab_bh = (int)tab[3];
    if (!DIRECT && hdr->overflow) return;  // (direct lists: the totals blocks of THIS launch decide that; the guards below keep every store inside its bin's segment)
    {
        uint32_t row0;
        gsr_view_rows(row_range, P, row0, P);  // with a row range P was only the capacity: records behind the view's last Gaussian were never written
    }
    if (i0 >= P) return;  // workgroup entirely behind the view's last Gaussian (before any barrier: uniform)
    // bin_offset of the table entries that hold instances: the one dependent gather
    const int area = tab_bw > 0 ? tab_bw * tab_bh : 0;
    uint32_t toff[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int t = k * SC_THREADS + (int)threadIdx.x;
        toff[k] = 0u;
        if (t < area && tent[k]) {
            const int ty = t / tab_bw, tx = t - ty * tab_bw;
            const int bin = (tab_by0 + ty) * bx + tab_bx0 + tx;
            toff[k] = DIRECT ? (uint32_t)bin * dd.bin_cap : bin_offset[bin];  // direct lists: the bin's segment starts at bin x capacity -- nothing to gather
        }
    }
    uint32_t lo[SC_PER], hi[SC_PER], mask[SC_PER];
    uint64_t key[SC_PER];
    GsrHit hit[SC_PER];
#pragma unroll
    for (int u = 0; u < SC_PER; u++) {
        const int i = i0 + u * SC_THREADS + (int)threadIdx.x;
        const bool in = i < P;
        lo[u] = in ? rec[u].y : 0u; hi[u] = in ? rec[u].z : 0u; mask[u] = in ? rec[u].w : 0u;
        key[u] = ((uint64_t)rec[u].x << 32) | (uint32_t)i;
        hit[u] = GsrHit{0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f};
        if ((hi[u] & 0xffff) > (lo[u] & 0xffff)) {  // listed somewhere
            const uint32_t ar = ((hi[u] & 0xffff) - (lo[u] & 0xffff)) * ((hi[u] >> 16) - (lo[u] >> 16));
            if (ar > 32u) {  // rect too large for the cached mask: the predicate k_preprocess counted with (row intervals, or per cell for
                             // ill-conditioned conics: gsr_masked_hit), re-evaluated from the splat record and the threshold k_preprocess left in the
                             // mask word (never recomputed here: gsr_hit_from_threshold)
                const float4 *sr = reinterpret_cast<const float4 *>(splats + i);
                const float4 a = sr[0], b = sr[1];
                hit[u] = gsr_hit_from_threshold(a.x, a.y, a.z, a.w, b.x, __uint_as_float(mask[u]));
            }
        }
    }
    if (tab_bw < 0) {
        // not recorded: this workgroup's bins did not fit the table in the count pass (incoherent input).  Rebuild -- half the Gaussians per call, any
        // split will do -- and reserve from the cursor the scan left behind the recorded instances of each bin.
#pragma unroll
        for (int u = 0; u < SC_PER; u++) {
            const uint64_t k64 = key[u];
            // scanned lists: the cursor starts behind the bin's recorded instances (k_scan_b); direct lists: it starts at 0 (zeroed with the counters) and
            // the recorded count is added here
            auto res = [&](int bin, uint32_t cnt) {
                const uint32_t c = atomicAdd(&bin_cursor[(size_t)bin * GSR_CPAD], cnt);
                return DIRECT ? (uint32_t)bin * dd.bin_cap + dd.bin_count[(size_t)bin * GSR_CPAD] + c : c;
            };
            gsr_block_bin<true, SC_THREADS>(s_cnt, s_base, s_box, lo[u], hi[u], bx, gsr_masked_hit(hit[u], mask[u], lo[u], hi[u]), res, res,
                                            [&](uint32_t pos, uint32_t, uint32_t bin) { if (!DIRECT || pos - bin * dd.bin_cap < dd.bin_cap) keys[pos] = k64; });
        }
        return;
    }
    if (area == 0) return;  // nothing listed in this workgroup (uniform)
    // slot of an instance = bin_offset[bin] + the base k_preprocess' count atomic returned for this workgroup + its rank here (LDS atomic)
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int t = k * SC_THREADS + (int)threadIdx.x;
        if (t < area) { s_cnt[t] = 0u; s_base[t] = toff[k] + tent[k] - 1u; }
    }
    for (int t = TPT * SC_THREADS + (int)threadIdx.x; t < area; t += SC_THREADS) {  // a box beyond the entries requested up front (uniform, rare)
        const uint32_t e = tab[4 + t];
        const int ty = t / tab_bw, tx = t - ty * tab_bw;
        s_cnt[t] = 0u;
        const int bin_ = (tab_by0 + ty) * bx + tab_bx0 + tx;
        s_base[t] = (e ? (DIRECT ? (uint32_t)bin_ * dd.bin_cap : bin_offset[bin_]) : 0u) + e - 1u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SC_PER; u++) {
        const uint64_t k64 = key[u];
        gsr_block_emit_one(s_cnt, s_base, tab_bx0, tab_by0, tab_bw, bx, lo[u], hi[u], gsr_masked_hit(hit[u], mask[u], lo[u], hi[u]),
                           [&](uint32_t pos, uint32_t, uint32_t bin) { if (!DIRECT || pos - bin * dd.bin_cap < dd.bin_cap) keys[pos] = k64; });  // (a list beyond the capacity: reported, not written)
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

// lists of 1..1024 keys (the common case: a body bin holds ~450): ONE WAVE per bin, keys in registers.
__global__ __launch_bounds__(64) void k_sort_wave(GsrBins bins, const uint32_t *__restrict__ wg_order, uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                                  const GsrHeader *__restrict__ hdr) {
    const uint32_t bin = wg_order[blockIdx.x];  // busy bins first (requested together with the header: one round trip, not two)
    if (hdr->overflow || blockIdx.x >= hdr->num_busy_wgs) return;
    uint32_t off, end;
    gsr_bin_range(bins, bin, off, end);
    const uint32_t n = end - off;
    if (n == 0 || n > 1024u) return;
    __shared__ uint32_t ids[1024];
    sort_wave_list(keys + off, n, point_list + off, ids, (int)threadIdx.x);
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

void gsr_launch_scan(const uint32_t *bin_count, const uint32_t *bin_count_fb, uint32_t *bin_offset, uint32_t *bin_cursor, uint32_t *wg_order, uint4 *scan_part, int NB, int bx, int by,
                     int64_t cap, GsrHeader *hdr, uint32_t *gpart, int n_gblocks, uint32_t *host_hdr, uint32_t host_seq, bool no_large_sort,
                     uint32_t order_hint, hipStream_t s) {
    const int NT = ((bx + 7) / 8) * ((by + 7) / 8) * 64;  // indices of the patch order (>= NB: ragged patches at the grid edge)
    const int nblocks = ((NT > NB ? NT : NB) + SB - 1) / SB;
    // (Tried: ONE 1,024-thread workgroup owning 16 indices per thread for grids of up to 16,384 bins -- no partials, no flags.  Bit-identical
    //  outputs, but 40.8 us against the fused form's 14.6: sixteen dependent rounds of patch-order index arithmetic and gathers on one CU cost far
    //  more than the release / acquire round trip they avoid.  Removed; profiles/r03_issue_probes.md section 4.)
    if (nblocks <= GSR_SCAN_FUSE_MAX) {
        hipLaunchKernelGGL(k_scan_b<true>, dim3(nblocks), dim3(SB), 0, s, bin_count, bin_count_fb, scan_part, bin_offset, bin_cursor, wg_order, NB, bx, by, nblocks, cap,
                           hdr, gpart, n_gblocks, host_hdr, host_seq, no_large_sort ? 1 : 0, order_hint);
        return;
    }
    hipLaunchKernelGGL(k_scan_a, dim3(nblocks), dim3(SB), 0, s, bin_count, bin_count_fb, scan_part, NB, bx, by, order_hint);
    hipLaunchKernelGGL(k_scan_b<false>, dim3(nblocks), dim3(SB), 0, s, bin_count, bin_count_fb, scan_part, bin_offset, bin_cursor, wg_order, NB, bx, by, nblocks, cap, hdr,
                       gpart, n_gblocks, host_hdr, host_seq, no_large_sort ? 1 : 0, order_hint);
}

void gsr_launch_scatter(int P, const uint32_t *row_range, int bx, const GsrSplat *splats, const uint4 *binrec, const uint32_t *wg_tab, const uint32_t *bin_offset,
                        uint32_t *bin_cursor, uint64_t *keys, const GsrHeader *hdr, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(k_scatter<false>, dim3((P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS), dim3(SC_THREADS), 0, s, P, row_range, bx, splats, binrec, wg_tab, bin_offset, bin_cursor, keys,
                       hdr, ScDirect{});
}

void gsr_launch_scatter_direct(int P, const uint32_t *row_range, int bx, int by, int NB, const GsrSplat *splats, const uint4 *binrec, const uint32_t *wg_tab, const uint32_t *bin_count,
                               const uint32_t *bin_count_fb, uint32_t *bin_cursor, uint32_t bin_cap, uint64_t *keys, GsrHeader *hdr, int64_t slot_cap, const uint32_t *gpart,
                               uint32_t *gprefix, int n_gblocks, uint32_t *wg_order, uint4 *scan_part, uint32_t order_hint, uint32_t *host_hdr, uint32_t host_seq, hipStream_t s) {
    if (P <= 0) return;
    const int NT = ((bx + 7) / 8) * ((by + 7) / 8) * 64;  // indices of the patch order (>= NB)
    ScDirect d;
    d.bin_count = bin_count; d.bin_count_fb = bin_count_fb; d.gpart = gpart; d.gprefix = gprefix; d.wg_order = wg_order;
    d.part = reinterpret_cast<unsigned long long *>(scan_part);
    d.hdr_w = hdr; d.host_hdr = host_hdr; d.slot_cap = slot_cap;
    d.bin_cap = bin_cap; d.host_seq = host_seq; d.hint = order_hint; d.NB = NB; d.n_gblocks = n_gblocks; d.by = by;
    d.nscan_wgs = ((NT > NB ? NT : NB) + SC_THREADS - 1) / SC_THREADS;
    hipLaunchKernelGGL(k_scatter<true>, dim3(SC_TOTALS + d.nscan_wgs + (P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS), dim3(SC_THREADS), 0, s, P, row_range, bx, splats, binrec, wg_tab,
                       nullptr, bin_cursor, keys, hdr, d);
}

void gsr_launch_sort(int NB, const uint32_t *bin_offset, const uint32_t *wg_order, uint64_t *keys, uint32_t *point_list,
                     const GsrHeader *hdr, bool no_large_sort, hipStream_t s) {
    if (NB <= 0) return;
    hipLaunchKernelGGL(k_sort_wave, dim3(NB), dim3(64), 0, s, GsrBins{bin_offset, nullptr, nullptr, 0u}, wg_order, keys, point_list, hdr);
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

void gsr_launch_sort_direct(int NB, GsrBins bins, const uint32_t *wg_order, uint64_t *keys, uint32_t *point_list, const GsrHeader *hdr, hipStream_t s) {
    if (NB <= 0) return;
    hipLaunchKernelGGL(k_sort_wave, dim3(NB), dim3(64), 0, s, bins, wg_order, keys, point_list, hdr);
}
