// capi.hip -- extern "C" entry points of libgpsgs_hip.so (declared in include/gpsgs.h).
// Pure enqueue: no allocation, no host synchronisation (except in the explicit *_read_header helper and under
// GSR_FLAG_DEBUG), so a forward+backward pair can be captured into a hipGraph.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "gsr_common.h"

namespace {

inline uint8_t *at(void *ws, size_t off) { return reinterpret_cast<uint8_t *>(ws) + off; }
inline const uint8_t *at(const void *ws, size_t off) { return reinterpret_cast<const uint8_t *>(ws) + off; }

// GPSGS_TRACE=1 (environment): every rasteriser stage announces itself on stderr before its launch and is synchronised after it --
// a device fault then names the stage that was running (a GPU memory fault aborts the process; there is nothing to return).
inline bool trace_on() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GPSGS_TRACE");
        v = (e && *e && *e != '0') ? 1 : 0;
    }
    return v == 1;
}
inline void trace(const char *what, int P, int W, int H, long long cap, unsigned flags) {
    if (trace_on()) { fprintf(stderr, "[gpsgs] %s P=%d %dx%d cap=%lld flags=0x%x ...", what, P, W, H, cap, flags); fflush(stderr); }
}

inline int check(hipStream_t s, unsigned flags) {
    if (hipGetLastError() != hipSuccess) return GPSGS_E_LAUNCH;
    if ((flags & GSR_FLAG_DEBUG) || trace_on()) {
        if (hipStreamSynchronize(s) != hipSuccess) return GPSGS_E_LAUNCH;
        if (hipGetLastError() != hipSuccess) return GPSGS_E_LAUNCH;
        if (trace_on()) { fprintf(stderr, " ok\n"); fflush(stderr); }
    }
    return GPSGS_OK;
}

// ---- optional per-stage hipEvent recorder (GSR_FLAG_TIMING) ---------------------------------------------------------
constexpr int TIMING_MAX = 8192;
struct Recorder {
    hipEvent_t ev[TIMING_MAX][2];
    int stage[TIMING_MAX];
    int created = 0, n = 0;
} g_rec;

struct StageTimer {  // RAII: records an event pair around one stage when timing is on
    bool on;
    int slot;
    hipStream_t s;
    StageTimer(unsigned flags, int stage, hipStream_t s_) : on((flags & GSR_FLAG_TIMING) != 0), slot(-1), s(s_) {
        const unsigned only = (flags >> 4) & 0xFu;  // GSR_FLAG_TIMING_STAGE(k): bracket stage k only (less perturbation)
        if (only && (int)only - 1 != stage) on = false;
        if (!on || g_rec.n >= TIMING_MAX) { on = false; return; }
        slot = g_rec.n++;
        if (slot >= g_rec.created) {
            (void)hipEventCreate(&g_rec.ev[slot][0]);
            (void)hipEventCreate(&g_rec.ev[slot][1]);
            g_rec.created = slot + 1;
        }
        g_rec.stage[slot] = stage;
        (void)hipEventRecord(g_rec.ev[slot][0], s);
    }
    ~StageTimer() {
        if (on) (void)hipEventRecord(g_rec.ev[slot][1], s);
    }
};

__global__ void k_export(int P, int T, const GsrSplat *__restrict__ splats, GsrBins bins, float *depth,
                         float *xy, float *conic_opacity, int *rect, int64_t *tile_ranges) {  // T = number of bins
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
        const GsrSplat s = splats[i];
        if (depth) depth[i] = s.depth;
        if (xy) { xy[2 * i] = s.x; xy[2 * i + 1] = s.y; }
        if (conic_opacity) { conic_opacity[4 * i] = s.A; conic_opacity[4 * i + 1] = s.B; conic_opacity[4 * i + 2] = s.C; conic_opacity[4 * i + 3] = s.op; }
        if (rect) { rect[4 * i] = s.bin_lo & 0xffff; rect[4 * i + 1] = s.bin_lo >> 16; rect[4 * i + 2] = s.bin_hi & 0xffff; rect[4 * i + 3] = s.bin_hi >> 16; }
    }
    if (i < T && tile_ranges) {
        uint32_t a, b;
        gsr_bin_range(bins, (uint32_t)i, a, b);
        tile_ranges[2 * i] = b > a ? (int64_t)a : 0;
        tile_ranges[2 * i + 1] = b > a ? (int64_t)b : 0;
    }
}

// GPSGS_TRACE only: after the sort, every entry of every bin list must be a Gaussian of the view and the lists must be in key order
__global__ void k_validate_lists(int P, int NB, GsrBins bins, const uint32_t *__restrict__ bin_cursor, const uint32_t *__restrict__ point_list,
                                 const GsrSplat *__restrict__ splats, unsigned long long *__restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= NB) return;
    uint32_t r0, r1;
    gsr_bin_range(bins, (uint32_t)b, r0, r1);
    // scanned lists: the scatter pass filled exactly the slots the count pass reserved (direct lists have no second pass: a slot nobody wrote shows
    // up below as an id that is not a Gaussian of the view, or out of order)
    if (!bins.cap && bin_cursor[(size_t)b * GSR_CPAD] != r1) atomicAdd(out + 4, 1ull);
    unsigned long long bad_id = 0, bad_order = 0, bad_rec = 0;
    uint64_t prev = 0;
    for (uint32_t k = r0; k < r1; k++) {
        const uint32_t id = point_list[k];
        if (id >= (uint32_t)P) { bad_id++; continue; }
        const GsrSplat s = splats[id];
        if (!(s.x == s.x) || !(s.A == s.A) || !(s.op == s.op)) bad_rec++;
        const uint64_t key = ((uint64_t)__float_as_uint(s.depth) << 32) | id;
        if (k > r0 && key <= prev) bad_order++;
        prev = key;
    }
    if (bad_id) atomicAdd(out + 0, bad_id);
    if (bad_order) atomicAdd(out + 1, bad_order);
    if (bad_rec) atomicAdd(out + 2, bad_rec);
    if (bad_id) atomicMax(out + 3, (unsigned long long)(r1 - r0));
}

// gsr_debug_count_records: how many gradient-record slots the last backward flagged (bench.py prices the record traffic with it)
__global__ __launch_bounds__(256) void k_count_flags(const uint8_t *__restrict__ flags, const GsrHeader *__restrict__ hdr, unsigned long long *__restrict__ out) {
    const uint32_t n = hdr->num_slots;
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) c += flags[i] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = n;
}

// the forward's counter section (header + scan partials + bin_count + bin_count_fb: 0.13 MB at 1024^2) zeroed by a plain kernel instead of the
// runtime's fill path: 4.3 us against 4.7 (min 2.5 / 2.6) in the round-5 kernel trace -- most of it is being the first launch behind the
// previous step's 160 MB of gradient-record traffic, not the zeroing
__global__ __launch_bounds__(256) void k_zero16(uint4 *__restrict__ p, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

extern "C" int gpsgs_abi_version(void) { return GPSGS_ABI_VERSION; }

extern "C" const char *gpsgs_build_info(void) { return "gfx950 hipcc " __VERSION__ " built " __DATE__; }

extern "C" size_t gsr_workspace_bytes(int P, int width, int height, int64_t instance_capacity) {
    if (P < 0 || width < 0 || height < 0 || instance_capacity < 0) return 0;
    return gsr_layout(P, width, height, instance_capacity).total;
}

extern "C" int gsr_direct_lists_ok(int width, int height, uint32_t bin_capacity) {
    if (width <= 0 || height <= 0 || bin_capacity == 0u || bin_capacity > GSR_DIRECT_MAX_CAP || (bin_capacity & 63u)) return 0;
    const GsrLayout L = gsr_layout(0, width, height, 0);
    const int nt = ((L.bx + 7) / 8) * ((L.by + 7) / 8) * 64;
    return (L.NB <= GSR_DIRECT_MAX_BINS && nt <= GSR_DIRECT_MAX_BINS) ? 1 : 0;
}

extern "C" size_t gsr_workspace_bytes_ex(int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity, int forward_only) {
    if (P < 0 || width < 0 || height < 0 || instance_capacity < 0) return 0;
    if (bin_capacity && !gsr_direct_lists_ok(width, height, bin_capacity)) return 0;
    const GsrLayout L = gsr_layout(P, width, height, instance_capacity, bin_capacity);
    return forward_only ? L.total_fwd : L.total;
}

extern "C" size_t gsr_workspace_bytes_forward_only(int P, int width, int height, int64_t instance_capacity) {
    if (P < 0 || width < 0 || height < 0 || instance_capacity < 0) return 0;
    return gsr_layout(P, width, height, instance_capacity).total_fwd;
}

extern "C" int gsr_forward_ex(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                              const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                              const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                              void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream,
                              void *host_header_out, uint32_t notify_seq, const GsrViewExt *ext) {
    const uint32_t *row_range = ext ? ext->row_range : nullptr;
    const uint32_t order_hint = ext ? ext->order_hint : 0u;
    if (P < 0 || width <= 0 || height <= 0 || instance_capacity < 0 || instance_capacity > 0x7fffffffLL) return GPSGS_E_INVALID;
    if (width > 65535 * GSR_TILE || height > 65535 * GSR_TILE) return GPSGS_E_INVALID;
    if (!out_color || !workspace) return GPSGS_E_INVALID;
    const float *shs = ext ? ext->shs : nullptr, *cov3D_precomp = ext ? ext->cov3D_precomp : nullptr;
    if (P > 0 && (!means3D || !opacities || !viewmatrix || !projmatrix || !bg || !radii)) return GPSGS_E_INVALID;
    // exactly one of (colors, shs) and exactly one of (scales + rotations, cov3D_precomp): upstream's GaussianRasterizer.forward raises on both
    if (P > 0 && ((shs != nullptr) == (colors != nullptr) || (cov3D_precomp != nullptr) == (scales != nullptr && rotations != nullptr))) return GPSGS_E_INVALID;
    if (P > 0 && cov3D_precomp && (scales || rotations)) return GPSGS_E_INVALID;
    if (shs && (!ext->campos || ext->sh_degree > 3u || ext->sh_coeffs > 16u || (ext->sh_degree + 1u) * (ext->sh_degree + 1u) > ext->sh_coeffs)) return GPSGS_E_INVALID;
    const uint32_t bin_cap = ext ? ext->bin_capacity : 0u;  // direct lists: a fixed-capacity segment per bin (GsrBins)
    if (bin_cap && !gsr_direct_lists_ok(width, height, bin_cap)) return GPSGS_E_INVALID;
    const GsrLayout L = gsr_layout(P, width, height, instance_capacity, bin_cap);
    if (workspace_bytes < L.total_fwd) return GPSGS_E_WORKSPACE;  // the backward tail is optional for a forward
    hipStream_t s = (hipStream_t)stream;
    uint32_t *host_hdr = nullptr;
    if (host_header_out) {  // must be pinned (hipHostMalloc / hipHostRegister) memory the device can write
        void *dptr = nullptr;
        if (notify_seq == 0 || hipHostGetDevicePointer(&dptr, host_header_out, 0) != hipSuccess || !dptr) {
            (void)hipGetLastError();
            return GPSGS_E_INVALID;
        }
        host_hdr = static_cast<uint32_t *>(dptr);
    }
    GsrHeader *hdr = reinterpret_cast<GsrHeader *>(at(workspace, L.header));
    uint32_t *bin_count = reinterpret_cast<uint32_t *>(at(workspace, L.bin_count));
    uint32_t *bin_count_fb = reinterpret_cast<uint32_t *>(at(workspace, L.bin_count_fb));
    uint32_t *bin_offset = reinterpret_cast<uint32_t *>(at(workspace, L.bin_offset));
    uint32_t *bin_cursor = reinterpret_cast<uint32_t *>(at(workspace, L.bin_cursor));
    uint32_t *wg_order = reinterpret_cast<uint32_t *>(at(workspace, L.wg_order));
    uint4 *scan_part = reinterpret_cast<uint4 *>(at(workspace, L.scan_part));
    GsrSplat *splats = reinterpret_cast<GsrSplat *>(at(workspace, L.splats));
    uint4 *binrec = reinterpret_cast<uint4 *>(at(workspace, L.binrec));
    uint32_t *wg_tab = reinterpret_cast<uint32_t *>(at(workspace, L.wg_tab));
    uint64_t *keys = reinterpret_cast<uint64_t *>(at(workspace, L.keys));
    uint32_t *point_list = reinterpret_cast<uint32_t *>(at(workspace, L.point_list));
    float *final_T = reinterpret_cast<float *>(at(workspace, L.final_T));
    uint32_t *n_contrib = reinterpret_cast<uint32_t *>(at(workspace, L.n_contrib));

    // header + scan partials (their ready flags) + bin_count + bin_count_fb (+ bin_cursor with direct lists, where no scan initialises it) are adjacent
    {
        const size_t zb = (bin_cap ? L.bin_offset : L.bin_cursor) - L.header;  // sections are 256-byte aligned: a whole number of 16-byte words
        if ((zb & 15u) == 0 && (reinterpret_cast<uintptr_t>(hdr) & 15u) == 0) {
            const uint32_t n16 = (uint32_t)(zb >> 4);
            hipLaunchKernelGGL(k_zero16, dim3((n16 + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<uint4 *>(hdr), n16);
            if (hipGetLastError() != hipSuccess) return GPSGS_E_LAUNCH;  // stale scan flags / bin counts would corrupt the whole view
        } else if (hipMemsetAsync(hdr, 0, zb, s) != hipSuccess) {
            return GPSGS_E_LAUNCH;
        }
    }
    if (P == 0) {  // upstream returns its zero-initialised image (NOT the background) when there is nothing to draw
        if (hipMemsetAsync(out_color, 0, sizeof(float) * 3 * (size_t)width * height, s) != hipSuccess) return GPSGS_E_LAUNCH;
        if (hipMemsetAsync(bin_offset, 0, (size_t)(L.NB + 1) * 4, s) != hipSuccess) return GPSGS_E_LAUNCH;
        return check(s, flags);
    }
    GsrFwdParams q;
    q.P = P; q.W = width; q.H = height; q.gx = L.gx; q.gy = L.gy; q.bx = L.bx; q.by = L.by; q.bx_real = L.bx_real;
    q.means3D = means3D; q.colors = colors; q.opacities = opacities; q.scales = scales; q.rotations = rotations;
    q.scale_modifier = scale_modifier; q.tanfovx = tanfovx; q.tanfovy = tanfovy;
    q.view = viewmatrix; q.proj = projmatrix; q.bg = bg; q.out_color = out_color; q.radii = radii; q.cap = instance_capacity;
    q.row_range = row_range;
    q.shs = shs; q.campos = shs ? ext->campos : nullptr; q.cov3D_precomp = cov3D_precomp;
    q.sh_degree = shs ? ext->sh_degree : 0u; q.sh_coeffs = shs ? ext->sh_coeffs : 0u;
    const GsrBins bins = {bin_offset, bin_count, bin_count_fb, bin_cap};
    // a workspace that includes the backward tail gets the per-Gaussian slot prefix and cleared record flags from the forward
    const bool training = workspace_bytes >= L.total;
    q.goff = training ? reinterpret_cast<uint32_t *>(at(workspace, L.goff)) : nullptr;
    q.gpart = training ? reinterpret_cast<uint32_t *>(at(workspace, L.gscan_part)) : nullptr;
    uint8_t *inst_valid_fwd = training ? reinterpret_cast<uint8_t *>(at(workspace, L.inst_valid)) : nullptr;
    const int n_gblocks = (P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS;

    // direct lists + tile family: the forward compositing waves sort their own lists (no sort launch).  Not with the debug self-check, which inspects the
    // sorted lists between the two; GPSGS_FUSED_SORT=0 keeps the separate launch (development: A/B timing)
    static const bool fused_sort_env = [] { const char *e = getenv("GPSGS_FUSED_SORT"); return !(e && e[0] == '0'); }();
    const bool fused_sort = bin_cap && (flags & GSR_FLAG_COMPOSITE_TILES) && fused_sort_env && !(trace_on() || (flags & GSR_FLAG_DEBUG));
    int rc;
    {
        trace("preprocess", P, width, height, (long long)instance_capacity, flags);
        StageTimer t(flags, GSR_STAGE_PREPROCESS, s);
        gsr_launch_preprocess(q, splats, binrec, wg_tab, bin_count, bin_count_fb, hdr, s);
    }
    if ((rc = check(s, flags)) != GPSGS_OK) return rc;
    if (bin_cap) {
        // direct lists: the scatter needs no offsets (slot = bin x capacity + recorded base + rank), its first workgroups publish the header (also towards the
        // host) and do what is left of the scan (work order, slot prefix) beside the scattering ones
        {
            trace("scatter (direct lists)", P, width, height, (long long)instance_capacity, flags);
            StageTimer t(flags, GSR_STAGE_SCATTER, s);
            gsr_launch_scatter_direct(P, row_range, L.bx, L.by, L.NB, splats, binrec, wg_tab, bin_count, bin_count_fb, bin_cursor, bin_cap, keys, hdr, instance_capacity, q.gpart,
                                      training ? reinterpret_cast<uint32_t *>(at(workspace, L.gprefix)) : nullptr, n_gblocks, wg_order, scan_part, order_hint, host_hdr, notify_seq, s);
        }
        if ((rc = check(s, flags)) != GPSGS_OK) return rc;
        if (!fused_sort) {
            trace("sort (direct lists)", P, width, height, (long long)instance_capacity, flags);
            StageTimer t(flags, GSR_STAGE_SORT, s);
            gsr_launch_sort_direct(L.NB, bins, wg_order, keys, point_list, hdr, s);
        }
    } else {
        {
            trace("scan", P, width, height, (long long)instance_capacity, flags);
            StageTimer t(flags, GSR_STAGE_SCAN, s);
            gsr_launch_scan(bin_count, bin_count_fb, bin_offset, bin_cursor, wg_order, scan_part, L.NB, L.bx, L.by, instance_capacity, hdr, q.gpart, n_gblocks, host_hdr, notify_seq,
                            (flags & GSR_FLAG_NO_LARGE_SORT) != 0, order_hint, s);
        }
        if ((rc = check(s, flags)) != GPSGS_OK) return rc;
        {
            trace("scatter", P, width, height, (long long)instance_capacity, flags);
            StageTimer t(flags, GSR_STAGE_SCATTER, s);
            gsr_launch_scatter(P, row_range, L.bx, splats, binrec, wg_tab, bin_offset, bin_cursor, keys, hdr, s);
        }
        if ((rc = check(s, flags)) != GPSGS_OK) return rc;
        {
            trace("sort", P, width, height, (long long)instance_capacity, flags);
            StageTimer t(flags, GSR_STAGE_SORT, s);
            gsr_launch_sort(L.NB, bin_offset, wg_order, keys, point_list, hdr, (flags & GSR_FLAG_NO_LARGE_SORT) != 0, s);
        }
    }
    if ((rc = check(s, flags)) != GPSGS_OK) return rc;
    if (trace_on() || (flags & GSR_FLAG_DEBUG)) {  // self-check between the sort and the compositing (synchronises; never in normal operation)
        // everything on the caller's stream `s` and checked: PyTorch's side streams do not synchronise with the NULL stream, so a NULL-stream memset of
        // the counters could land AFTER k_validate_lists' atomics and hide a real inconsistency.  The five counters live in a small
        // device buffer allocated once per process (a debug forward is never captured, but it no longer allocates per call either).
        GsrHeader h;
        unsigned long long h_out[5] = {0, 0, 0, 0, 0};
        static unsigned long long *d_out_by_dev[64] = {nullptr};
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return GPSGS_E_LAUNCH;
        unsigned long long *d_out;
        {
            static std::mutex mu;
            std::lock_guard<std::mutex> g(mu);
            if (!d_out_by_dev[devid] && hipMalloc(&d_out_by_dev[devid], sizeof(h_out)) != hipSuccess) return GPSGS_E_LAUNCH;
            d_out = d_out_by_dev[devid];
        }
        if (hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipMemsetAsync(d_out, 0, sizeof(h_out), s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return GPSGS_E_LAUNCH;
        if (!h.overflow) hipLaunchKernelGGL(k_validate_lists, dim3((L.NB + 255) / 256), dim3(256), 0, s, P, L.NB, bins, bin_cursor, point_list, splats, d_out);
        if (hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return GPSGS_E_LAUNCH;
        const bool bad = h_out[0] || h_out[1] || h_out[4];
        if (trace_on() || bad) {
            fprintf(stderr, "[gpsgs] header: R=%llu overflow=%u longest=%u busy=%u slots=%u points=%u | lists: ids out of range %llu, out of order %llu, non-finite records %llu, "
                            "bins whose scatter cursor missed its end %llu (longest list with a bad id %llu)\n",
                    (unsigned long long)h.num_rendered, h.overflow, h.max_tile_count, h.num_busy_wgs, h.num_slots, h.num_points, h_out[0], h_out[1], h_out[2], h_out[4], h_out[3]);
            fflush(stderr);
        }
        if (bad) return GPSGS_E_INTERNAL;
    }
    {
        trace("composite_fwd", P, width, height, (long long)instance_capacity, flags);
        StageTimer t(flags, GSR_STAGE_COMPOSITE_FWD, s);
        if (flags & GSR_FLAG_COMPOSITE_TILES)
            gsr_launch_composite_fwd_tiles(width, height, L.bx, L.by, splats, bins, wg_order, point_list, bg, out_color, final_T, n_contrib, hdr, inst_valid_fwd, training,
                                           (flags & GSR_FLAG_WAVE_PRIORITY) != 0, fused_sort ? keys : nullptr, s);
        else
            gsr_launch_composite_fwd(width, height, L.bx, L.by, splats, bins, wg_order, point_list, bg, out_color, final_T, n_contrib, hdr, inst_valid_fwd, s);
    }
    return check(s, flags);
}

extern "C" int gsr_forward_notify(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                                  const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                                  const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                                  void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream,
                                  void *host_header_out, uint32_t notify_seq) {
    return gsr_forward_ex(P, width, height, means3D, colors, opacities, scales, rotations, scale_modifier, tanfovx, tanfovy, viewmatrix, projmatrix,
                          bg, out_color, radii, workspace, workspace_bytes, instance_capacity, flags, stream, host_header_out, notify_seq, nullptr);
}

extern "C" int gsr_forward(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                           const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                           const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                           void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream) {
    return gsr_forward_ex(P, width, height, means3D, colors, opacities, scales, rotations, scale_modifier, tanfovx, tanfovy, viewmatrix, projmatrix,
                          bg, out_color, radii, workspace, workspace_bytes, instance_capacity, flags, stream, nullptr, 0u, nullptr);
}

extern "C" int gsr_backward_ex(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                               const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                               const float *viewmatrix, const float *projmatrix, const float *bg, const int *radii,
                               const float *dL_dpix, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                               float *dL_dscales, float *dL_drotations, void *workspace, size_t workspace_bytes,
                               int64_t instance_capacity, unsigned flags, void *stream, const GsrViewExt *ext) {
    (void)colors; (void)opacities;  // already folded into the splat records of the workspace
    if (P < 0 || width <= 0 || height <= 0 || instance_capacity < 0) return GPSGS_E_INVALID;
    if (P == 0) return GPSGS_OK;
    const float *shs = ext ? ext->shs : nullptr, *cov3D_precomp = ext ? ext->cov3D_precomp : nullptr;
    if (!means3D || !viewmatrix || !projmatrix || !bg || !radii || !dL_dpix || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity || !workspace) return GPSGS_E_INVALID;
    if (!shs && !dL_dcolors) return GPSGS_E_INVALID;
    if (cov3D_precomp ? (!ext->dL_dcov3D || scales || rotations) : (!scales || !rotations || !dL_dscales || !dL_drotations)) return GPSGS_E_INVALID;
    if (shs && (!ext->campos || !ext->dL_dsh || ext->sh_degree > 3u || ext->sh_coeffs > 16u || (ext->sh_degree + 1u) * (ext->sh_degree + 1u) > ext->sh_coeffs))
        return GPSGS_E_INVALID;
    const uint32_t bin_cap = ext ? ext->bin_capacity : 0u;  // must be what the forward was given
    if (bin_cap && !gsr_direct_lists_ok(width, height, bin_cap)) return GPSGS_E_INVALID;
    const GsrLayout L = gsr_layout(P, width, height, instance_capacity, bin_cap);
    if (workspace_bytes < L.total) return GPSGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const GsrHeader *hdr = reinterpret_cast<const GsrHeader *>(at(workspace, L.header));
    const GsrBins bins = {reinterpret_cast<const uint32_t *>(at(workspace, L.bin_offset)), reinterpret_cast<const uint32_t *>(at(workspace, L.bin_count)),
                          reinterpret_cast<const uint32_t *>(at(workspace, L.bin_count_fb)), bin_cap};
    const uint32_t *wg_order = reinterpret_cast<const uint32_t *>(at(workspace, L.wg_order));
    const GsrSplat *splats = reinterpret_cast<const GsrSplat *>(at(workspace, L.splats));
    const uint32_t *point_list = reinterpret_cast<const uint32_t *>(at(workspace, L.point_list));
    const float *final_T = reinterpret_cast<const float *>(at(workspace, L.final_T));
    const uint32_t *n_contrib = reinterpret_cast<const uint32_t *>(at(workspace, L.n_contrib));
    uint32_t *goff = reinterpret_cast<uint32_t *>(at(workspace, L.goff));
    uint32_t *gscan_part = reinterpret_cast<uint32_t *>(at(workspace, L.gprefix));  // (the slot prefix per binning workgroup)
    uint8_t *inst_valid = reinterpret_cast<uint8_t *>(at(workspace, L.inst_valid));
    float *inst_dop = reinterpret_cast<float *>(at(workspace, L.inst_dop));
    GsrGradAcc *inst_grad = reinterpret_cast<GsrGradAcc *>(at(workspace, L.inst_grad));

    // goff / gscan_part / cleared inst_valid were produced by the matching gsr_forward (training workspace)
    int rc;
    {
        trace("composite_bwd", P, width, height, (long long)instance_capacity, flags);
        StageTimer t(flags, GSR_STAGE_COMPOSITE_BWD, s);
        // must be the same family as the forward that filled the workspace: the two designs round the exponent differently, and the
        // backward has to repeat the forward's per-pixel decisions
        if (flags & GSR_FLAG_COMPOSITE_TILES)
            gsr_launch_composite_bwd_tiles(width, height, L.bx, L.by, splats, bins, wg_order, point_list, bg, dL_dpix, final_T, n_contrib, goff, gscan_part, inst_valid, inst_dop, inst_grad, hdr,
                                           (flags & GSR_FLAG_NO_COLOR_GRAD) == 0, (flags & GSR_FLAG_WAVE_PRIORITY) != 0, s);
        else
            gsr_launch_composite_bwd(width, height, L.bx, L.by, splats, bins, wg_order, point_list, bg, dL_dpix, final_T, n_contrib, goff, gscan_part, inst_valid, inst_dop, inst_grad, hdr, s);
    }
    if ((rc = check(s, flags)) != GPSGS_OK) return rc;
    GsrBwdParams b;
    b.P = P; b.W = width; b.H = height;
    b.means3D = means3D; b.scales = scales; b.rotations = rotations;
    b.scale_modifier = scale_modifier; b.tanfovx = tanfovx; b.tanfovy = tanfovy;
    b.view = viewmatrix; b.proj = projmatrix; b.radii = radii;
    b.dL_dmeans3D = dL_dmeans3D; b.dL_dmeans2D = dL_dmeans2D; b.dL_dcolors = dL_dcolors; b.dL_dopacity = dL_dopacity;
    b.dL_dscales = dL_dscales; b.dL_drotations = dL_drotations;
    b.row_range = ext ? ext->row_range : nullptr;
    b.shs = shs; b.campos = shs ? ext->campos : nullptr; b.cov3D_precomp = cov3D_precomp;
    b.sh_degree = shs ? ext->sh_degree : 0u; b.sh_coeffs = shs ? ext->sh_coeffs : 0u;
    b.dL_dsh = shs ? ext->dL_dsh : nullptr; b.dL_dcov3D = cov3D_precomp ? ext->dL_dcov3D : nullptr;
    b.dop_in_record = ((flags & GSR_FLAG_COMPOSITE_TILES) && (flags & GSR_FLAG_NO_COLOR_GRAD)) ? 1 : 0;  // what k_composite_bwd_tiles<false> just wrote
    {
        trace("preprocess_bwd", P, width, height, (long long)instance_capacity, flags);
        StageTimer t(flags, GSR_STAGE_PREPROCESS_BWD, s);
        gsr_launch_preprocess_bwd(b, splats, goff, gscan_part, inst_valid, inst_dop, inst_grad, hdr, s);
    }
    return check(s, flags);
}

extern "C" int gsr_backward(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                            const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                            const float *viewmatrix, const float *projmatrix, const float *bg, const int *radii,
                            const float *dL_dpix, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                            float *dL_dscales, float *dL_drotations, void *workspace, size_t workspace_bytes,
                            int64_t instance_capacity, unsigned flags, void *stream) {
    return gsr_backward_ex(P, width, height, means3D, colors, opacities, scales, rotations, scale_modifier, tanfovx, tanfovy, viewmatrix, projmatrix, bg,
                           radii, dL_dpix, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, workspace, workspace_bytes,
                           instance_capacity, flags, stream, nullptr);
}

extern "C" int gsr_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present, void *stream) {
    (void)projmatrix;  // upstream's in_frustum() projects the point too, but tests only the view-space depth
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return GPSGS_E_INVALID;
    gsr_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int gsr_selftest(float *out4_device, void *stream) {
    if (!out4_device) return GPSGS_E_INVALID;
    gsr_launch_selftest(out4_device, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

// ---- shader-clock measurement (diagnostic): s_memtime ticks once per shader cycle, wall_clock64() at the constant rate the runtime reports
// (hipDeviceAttributeWallClockRate); a chip-filling grid of dependent FMAs runs for a few milliseconds and wave 0 brackets itself with both.
namespace {
__global__ __launch_bounds__(256) void k_sclk(unsigned long long *out, int iters) {
    float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) a = __builtin_fmaf(a, b, 1e-7f);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (a == 123.456f) out[2] = 1ull;  // keeps the loop alive
}
}  // namespace

extern "C" int gpsgs_measure_sclk(unsigned long long *scratch3_device, double *mhz_host, void *stream) {
    if (!scratch3_device || !mhz_host) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, rate_khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0)
        return GPSGS_E_NO_DEVICE;
    hipLaunchKernelGGL(k_sclk, dim3(2048), dim3(256), 0, s, scratch3_device, 60000);
    unsigned long long h[2] = {0ull, 0ull};
    if (hipMemcpyAsync(h, scratch3_device, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess) return GPSGS_E_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return GPSGS_E_LAUNCH;
    if (h[1] == 0ull) return GPSGS_E_LAUNCH;
    *mhz_host = (double)h[0] / (double)h[1] * (double)rate_khz * 1e-3;
    return GPSGS_OK;
}

extern "C" int gsr_debug_set_wg_trace(unsigned long long *rows_device) { return gsr_set_wg_trace(rows_device) == 0 ? GPSGS_OK : GPSGS_E_LAUNCH; }

extern "C" int gsr_debug_count_records(const void *workspace, size_t workspace_bytes, int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity,
                                       unsigned long long *out2_device, void *stream) {
    if (!workspace || !out2_device || P < 0 || width <= 0 || height <= 0 || instance_capacity < 0) return GPSGS_E_INVALID;
    if (bin_capacity && !gsr_direct_lists_ok(width, height, bin_capacity)) return GPSGS_E_INVALID;
    const GsrLayout L = gsr_layout(P, width, height, instance_capacity, bin_capacity);
    if (workspace_bytes < L.total) return GPSGS_E_WORKSPACE;  // a forward-only workspace has no record flags (they lie beyond its end)
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out2_device, 0, 16, s) != hipSuccess) return GPSGS_E_LAUNCH;
    hipLaunchKernelGGL(k_count_flags, dim3(1024), dim3(256), 0, s, at(workspace, L.inst_valid), reinterpret_cast<const GsrHeader *>(at(workspace, L.header)), out2_device);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int gsr_timing_read(float *ms_sum_host, int *launches_host) {
    if (!ms_sum_host || !launches_host) return GPSGS_E_INVALID;
    for (int i = 0; i < GSR_STAGE_COUNT; i++) { ms_sum_host[i] = 0.f; launches_host[i] = 0; }
    if (hipDeviceSynchronize() != hipSuccess) return GPSGS_E_LAUNCH;
    for (int i = 0; i < g_rec.n; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_rec.ev[i][0], g_rec.ev[i][1]) != hipSuccess) continue;
        ms_sum_host[g_rec.stage[i]] += ms;
        launches_host[g_rec.stage[i]] += 1;
    }
    g_rec.n = 0;
    return GPSGS_OK;
}

extern "C" int gsr_copy_header_async(const void *workspace, void *host_pinned_out, void *stream) {
    if (!workspace || !host_pinned_out) return GPSGS_E_INVALID;
    return hipMemcpyAsync(host_pinned_out, workspace, 32, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}

extern "C" int gsr_read_header(const void *workspace, GsrHeader *host_out, void *stream) {
    if (!workspace || !host_out) return GPSGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(host_out, workspace, sizeof(GsrHeader), hipMemcpyDeviceToHost, s) != hipSuccess) return GPSGS_E_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return GPSGS_E_LAUNCH;
    return GPSGS_OK;
}

extern "C" int gsr_export_state(const void *workspace, int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity, float *depth, float *xy,
                                float *conic_opacity, int *rect, int64_t *tile_ranges, uint32_t *point_list, float *final_T,
                                uint32_t *n_contrib, void *stream) {
    if (!workspace || P < 0 || width <= 0 || height <= 0) return GPSGS_E_INVALID;
    if (bin_capacity && !gsr_direct_lists_ok(width, height, bin_capacity)) return GPSGS_E_INVALID;
    const GsrLayout L = gsr_layout(P, width, height, instance_capacity, bin_capacity);
    hipStream_t s = (hipStream_t)stream;
    const int n = P > L.NB ? P : L.NB;
    const GsrBins bins = {reinterpret_cast<const uint32_t *>(at(workspace, L.bin_offset)), reinterpret_cast<const uint32_t *>(at(workspace, L.bin_count)),
                          reinterpret_cast<const uint32_t *>(at(workspace, L.bin_count_fb)), bin_capacity};
    if (n > 0)
        hipLaunchKernelGGL(k_export, dim3((n + 255) / 256), dim3(256), 0, s, P, L.NB, reinterpret_cast<const GsrSplat *>(at(workspace, L.splats)), bins, depth, xy,
                           conic_opacity, rect, tile_ranges);
    const size_t npix = (size_t)width * height;
    if (point_list && L.key_cap > 0)  // point_list [key capacity]: the instance capacity, or bins x bin_capacity (direct lists: every bin's segment, gaps included)
        (void)hipMemcpyAsync(point_list, at(workspace, L.point_list), (size_t)L.key_cap * 4, hipMemcpyDeviceToDevice, s);
    if (final_T) (void)hipMemcpyAsync(final_T, at(workspace, L.final_T), npix * 4, hipMemcpyDeviceToDevice, s);
    if (n_contrib) (void)hipMemcpyAsync(n_contrib, at(workspace, L.n_contrib), npix * 4, hipMemcpyDeviceToDevice, s);
    return hipGetLastError() == hipSuccess ? GPSGS_OK : GPSGS_E_LAUNCH;
}
