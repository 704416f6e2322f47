// gsr_sort_wave.h -- one bin list of 1..1024 keys sorted by ONE wave with the keys in registers (sort_wave_list), and the lane-exchange / key-order
// helpers the multi-wave sort kernels share with it.  Included by gsr_binning.hip (k_sort_wave, k_sort_multi) and by gsr_composite_tiles.hip (direct
// lists: the forward compositing wave sorts its own bin's list before it walks it -- no sort launch).
#pragma once
#include "gsr_common.h"

namespace {

__device__ __forceinline__ void sort_sync(bool wave_only) {
    if (wave_only) {  // a single wave owns the segment: the DS queue is in order, only the compiler needs pinning
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
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
__device__ __forceinline__ void sort_wave_regs(const uint64_t *__restrict__ seg, uint32_t n, uint32_t *__restrict__ out, uint32_t *ids, int lane) {
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
        if (e < n) { out[e] = (uint32_t)key[r]; ids[e] = (uint32_t)key[r]; }
    }
}

// ---- the same network on 32-bit COMPOSITE keys (round 5) ------------------------------------------------------------------------------
// The 64-bit form above costs a lane exchange 4 cross-lane moves + v_min_f64 + v_max_f64 (half rate) + 4 selects per key; the sort kernel is
// instruction-issue bound (round-4 counters).  Inside ONE bin the depths span a narrow range of float bit patterns (a body bin of config 2: a few
// 10^5 of the 2^32), so (depth bits - smallest depth bits of the bin) fits the upper 32 - L bits of a word whose lower L bits hold the key's
// POSITION in the unsorted segment (L = bits of n - 1 <= 10): one cross-lane move + v_min_u32 + v_max_u32 + one select per key, everything at full
// rate.  Order of distinct depths = order of the composites, exactly.  Equal depths must come out in ascending Gaussian id (upstream's stable
// order), which the position bits do not give: after the sort the ids of neighbours with equal depth fields are put in order by a short
// odd-even transposition (sort_wave_regs32).  A list whose depth range does not fit is handed to the exact 64-bit sort instead.  Ids wait in
// LDS (4 bytes per key) and are fetched by position once the order is known.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    __asm__("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int KPL, int KB, int ST>
__device__ __forceinline__ void sort_stage32(uint32_t (&key)[KPL], int lane) {
    constexpr int LOGK = KPL == 1 ? 0 : KPL == 2 ? 1 : KPL == 4 ? 2 : KPL == 8 ? 3 : 4;
    constexpr uint32_t mask = ST == 0 ? ((1u << KB) - 1u) : (1u << (KB - 1 - ST));
    constexpr uint32_t top = ST == 0 ? (1u << (KB - 1)) : mask;
    constexpr uint32_t rmask = mask & (KPL - 1), lmask = mask >> LOGK;
    if constexpr (lmask == 0) {
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int r2 = r ^ (int)rmask;
            if (r2 > r) {
                const uint32_t a = key[r], b = key[r2];
                key[r] = min(a, b);
                key[r2] = max(a, b);
            }
        }
    } else {
        // the lower element of a pair keeps the minimum, the upper one the maximum: median(a, b, 0) = min(a, b), median(a, b, ~0) = max(a, b) -- ONE v_med3_u32
        // per key with the lane's role as a per-stage constant, instead of v_min + v_max + a select (round 6: 4 -> 2 instructions per exchanged key, the
        // cross-lane move included; 21 of the 45 stages of a 512-key sort are of this kind)
        const uint32_t role = ((uint32_t)lane & (top >> LOGK)) == 0 ? 0u : 0xffffffffu;
        uint32_t other[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) other[r] = lane_xor<lmask>(key[r ^ (int)rmask]);
#pragma unroll
        for (int r = 0; r < KPL; r++) key[r] = umed3(key[r], other[r], role);
    }
}
template <int KPL, int LOGN, int KB, int ST>
__device__ __forceinline__ void sort_stages_from32(uint32_t (&key)[KPL], int lane) {
    sort_stage32<KPL, KB, ST>(key, lane);
    if constexpr (ST + 1 < KB) sort_stages_from32<KPL, LOGN, KB, ST + 1>(key, lane);
    else if constexpr (KB < LOGN) sort_stages_from32<KPL, LOGN, KB + 1, 0>(key, lane);
}

// -> true: `out` holds the sorted ids; false: this list needs the 64-bit sort (depth range too wide for the composite)
template <int KPL>
__device__ __forceinline__ bool sort_wave_regs32(const uint64_t *__restrict__ seg, uint32_t n, uint32_t *__restrict__ out, uint32_t *ids, int lane) {
    constexpr int LOGK = KPL == 1 ? 0 : KPL == 2 ? 1 : KPL == 4 ? 2 : KPL == 8 ? 3 : 4;
    constexpr int LOGN = LOGK + 6;
    uint32_t key[KPL], lo[KPL];
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    // (keys arrive unsorted: ANY assignment to the network's elements will do, so they are read COALESCED -- register r of lane l takes key
    //  r * 64 + l -- and that index is the position the composite carries.  Round-5 counters: with lane-strided loads and stores this kernel issued
    //  4x / 8x the memory requests its bytes need and spent 60 % of its wave cycles waiting for them; the network itself is a quarter of its time.)
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
        const uint64_t k = e < n ? seg[e] : 0ull;
        key[r] = (uint32_t)(k >> 32);
        lo[r] = (uint32_t)k;
        if (e < n) { dmin = min(dmin, key[r]); dmax = max(dmax, key[r]); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, 64));
    }
    const uint32_t L = 32u - (uint32_t)__builtin_clz(n - 1u);  // n >= 2: 2^L >= n
    if (!(dmax - dmin < (0xffffffffu >> L))) return false;     // (wave-uniform) the composite of a real key stays below the padding value
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
        if (e < n) ids[e] = lo[r];
        key[r] = e < n ? (((key[r] - dmin) << L) | e) : 0xffffffffu;
    }
    sort_stages_from32<KPL, LOGN, 1, 0>(key, lane);
    // ids by position (element e = lane * KPL + r now holds the e-th smallest composite)
    sort_sync(true);  // the ids written above are read by other lanes of this wave
    const uint32_t pm = (1u << L) - 1u;
    uint32_t idr[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) idr[r] = key[r] != 0xffffffffu ? ids[key[r] & pm] : 0xffffffffu;
    // EQUAL DEPTHS must come out in ascending Gaussian id; the position bits ordered them arbitrarily.  Inside a bin the depths are so close that a
    // few hundred keys usually hold a tie or two (config 2: ~40,000 distinct bit patterns per bin), so this is the normal case, not an exception:
    // odd-even transposition of the IDS over neighbours with equal depth fields (the runs are 2-3 elements long), until a pass swaps nothing.
    uint32_t eq = 0u;  // bit r: element (lane, r) and its successor in element order carry the same depth
    {
        const uint32_t next0 = (uint32_t)__shfl_down((int)key[0], 1, 64);
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint32_t nx = r + 1 < KPL ? key[r + 1 < KPL ? r + 1 : r] : (lane < 63 ? next0 : 0xffffffffu);
            eq |= (nx != 0xffffffffu && (nx >> L) == (key[r] >> L)) ? (1u << r) : 0u;
        }
    }
    if (__ballot(eq != 0u) != 0ull) {
        // (the shuffle runs with EVERY lane active and the lane-0 case is selected afterwards: inside `lane > 0 ? shuffle : 0` lane 1 would read an
        //  inactive lane 0 -- that was a real bug: duplicated ids at elements 15 / 16 of 11 of 20,331 lists of config 5)
        const uint32_t eq_up = (uint32_t)__shfl_up((int)eq, 1, 64);
        const uint32_t eq_prev = lane > 0 ? (eq_up >> (KPL - 1)) & 1u : 0u;  // my first element continues the previous lane's last run
        for (;;) {
            bool sw = false;
#pragma unroll
            for (int ph = 0; ph < 2; ph++) {  // pairs (e, e + 1) with e even, then with e odd; e = lane * KPL + r
#pragma unroll
                for (int r = 0; r + 1 < KPL; r++) {
                    if ((r & 1) != ph) continue;  // (KPL is even here: the parity of e is the parity of r)
                    const bool x = ((eq >> r) & 1u) && idr[r] > idr[r + 1];
                    const uint32_t a = idr[r], b2 = idr[r + 1];
                    idr[r] = x ? b2 : a;
                    idr[r + 1] = x ? a : b2;
                    sw |= x;
                }
                // the pair that straddles two lanes: (lane, KPL - 1) <-> (lane + 1, 0); its e is odd for KPL > 1, lane-parity dependent for KPL == 1
                if (KPL > 1 ? ph == 1 : true) {
                    const bool mine = KPL > 1 || (lane & 1) == ph;          // I hold the pair's first element in this phase
                    const bool theirs = KPL > 1 || ((lane - 1) & 1) == ph;  // ... my predecessor does, and I hold the second
                    const uint32_t nb = (uint32_t)__shfl_down((int)idr[0], 1, 64);        // successor lane's first id
                    const uint32_t pa = (uint32_t)__shfl_up((int)idr[KPL - 1], 1, 64);    // predecessor lane's last id
                    const bool x1 = mine && ((eq >> (KPL - 1)) & 1u) && idr[KPL - 1] > nb;
                    const bool x0 = theirs && eq_prev && pa > idr[0];
                    if (KPL > 1) {
                        idr[KPL - 1] = x1 ? nb : idr[KPL - 1];
                        idr[0] = x0 ? pa : idr[0];
                    } else {
                        idr[0] = x1 ? nb : (x0 ? pa : idr[0]);
                    }
                    sw |= x1;
                }
            }
            if (__ballot(sw) == 0ull) break;
        }
    }
    // the sorted ids leave through LDS once more: element lane * KPL + r sits in lane `lane`, a coalesced store wants element r * 64 + lane
    sort_sync(true);  // (every lane has fetched its ids)
#pragma unroll
    for (int r = 0; r < KPL; r++) ids[(uint32_t)lane * KPL + r] = idr[r];
    sort_sync(true);
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
        if (e < n) out[e] = ids[e];
    }
    return true;
}

// one list of 1..1024 keys sorted by ONE wave, keys in registers (ids: 1,024 words of LDS).  On return out[0, n) AND ids[0, n) hold the sorted ids (a caller
// that goes on to read ids[] synchronises the wave first: sort_sync(true)).
__device__ __forceinline__ void sort_wave_list(const uint64_t *__restrict__ seg, uint32_t n, uint32_t *__restrict__ out, uint32_t *ids, int lane) {
    if (n == 1u) { if (lane == 0) { const uint32_t id = (uint32_t)seg[0]; out[0] = id; ids[0] = id; } return; }
    bool done;
    if (n <= 64u) done = sort_wave_regs32<1>(seg, n, out, ids, lane);
    else if (n <= 128u) done = sort_wave_regs32<2>(seg, n, out, ids, lane);
    else if (n <= 256u) done = sort_wave_regs32<4>(seg, n, out, ids, lane);
    else if (n <= 512u) done = sort_wave_regs32<8>(seg, n, out, ids, lane);
    else done = sort_wave_regs32<16>(seg, n, out, ids, lane);
    if (done) return;  // (wave-uniform; otherwise: the exact 64-bit network below)
    if (n <= 64u) sort_wave_regs<1>(seg, n, out, ids, lane);
    else if (n <= 128u) sort_wave_regs<2>(seg, n, out, ids, lane);
    else if (n <= 256u) sort_wave_regs<4>(seg, n, out, ids, lane);
    else if (n <= 512u) sort_wave_regs<8>(seg, n, out, ids, lane);
    else sort_wave_regs<16>(seg, n, out, ids, lane);
}

}  // namespace
