/*
 * gpsgs.h -- C ABI of the MI355X-native render hot path of GPS-Gaussian (libgpsgs_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types, no exceptions, no allocation and no
 * host synchronisation inside any entry point (all are hipGraph-capture safe).  Every function returns 0 on
 * success or a negative GPSGS_E_* code.  All pointers are DEVICE pointers unless marked host.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   gsr_forward / gsr_backward
 *       the `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` calls made by the external CUDA
 *       extension `diff_gaussian_rasterization`, which the reference imports at gaussian_renderer/__init__.py:14
 *       and invokes at gaussian_renderer/__init__.py:51-62 (GaussianRasterizer.forward) and, through autograd,
 *       from train_stage2.py:83 (backward).  Argument meaning follows that call: precomputed colours, scale + rotation.  The other
 *       half of that module's interface -- `shs` (settings fields sh_degree / campos, constructed at gaussian_renderer/__init__.py:46-47)
 *       and `cov3D_precomp`, both passed as None by the reference (:56,:61) -- travels in GsrViewExt.
 *   cs_forward / cs_backward
 *       `corr_sampler.forward` / `corr_sampler.backward` of the external RAFT-Stereo sampler extension, called at
 *       core/corr.py:22 and core/corr.py:28.
 *   cv_build_forward / cv_build_backward, cs_lookup_forward / cs_lookup_backward
 *       CorrBlockFast1D: the all-pairs correlation volume + average-pool pyramid (core/corr.py:31-42, :53-61) and the lookup of
 *       all levels with the concatenation of core/corr.py:44-51 (four corr_sampler.forward calls + torch.cat in the reference).
 *   cu_upsample_forward / cu_upsample_backward
 *       RAFTStereoHuman.upsample_flow, core/raft_stereo_human.py:69-81 (softmax over the 9 taps + unfold + weighted sum).
 *   gsr_pack_views / gsr_pack_views_backward
 *       the per-sample flatten + boolean-mask gather + concat + rgb affine of lib/GaussianRender.py:15-34.
 *   up_unproject_forward / up_unproject_backward (+ _dev: cameras in device memory)
 *       flow2depth + depth2pc + the validity test: lib/utils.py:113-120, :88-110, lib/network.py:66-69.
 *   fl_l1_ssim_forward / fl_l1_ssim_backward
 *       l1_loss + ssim of lib/loss.py:36-83 (and their autograd backward), called at train_stage2.py:70-72.
 */
#ifndef GPSGS_H
#define GPSGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSGS_ABI_VERSION 4 /* 2: + GsrViewExt, gsr_forward_ex, gsr_backward_ex; header words num_points / row_overflow (additive)
                               3: GsrViewExt grew from 32 to 80 bytes: SH colours and precomputed 3D covariances (the `shs` / `cov3D_precomp`
                                  inputs of the upstream module, + their gradients); zero-initialised it means what version 2 meant
                               4: + gsr_mark_visible (upstream GaussianRasterizer.markVisible); gsr_debug_count_records takes workspace_bytes;
                                  DIRECT bin lists: GsrViewExt.reserved0 became bin_capacity (0 = what version 3 did), gsr_workspace_bytes_ex,
                                  gsr_direct_lists_ok; gsr_export_state / gsr_debug_count_records take bin_capacity */

enum {
    GPSGS_OK = 0,
    GPSGS_E_INVALID = -1,    /* bad argument (NULL pointer, negative size, unsupported dtype) */
    GPSGS_E_WORKSPACE = -2,  /* workspace_bytes smaller than gsr_workspace_bytes() for these dimensions */
    GPSGS_E_LAUNCH = -3,     /* a HIP launch failed (hipGetLastError != hipSuccess) */
    GPSGS_E_NO_DEVICE = -4,  /* no HIP device */
    GPSGS_E_INTERNAL = -5    /* GSR_FLAG_DEBUG only: the forward's self-check failed (a bin list holds an id that is not a Gaussian of the view, is
                                out of (depth, index) order, or the scatter pass did not fill exactly the slots the count pass reserved) */
};

/* flags for gsr_forward / gsr_backward */
#define GSR_FLAG_DEBUG 1u  /* synchronise and check after every kernel (the reference hard-codes debug=False); the forward also validates its
                              bin lists between the sort and the compositing (GPSGS_E_INTERNAL) */
#define GSR_FLAG_TIMING 2u /* bracket every stage with hipEvents on `stream`; read them with gsr_timing_read() */
#define GSR_FLAG_NO_LARGE_SORT 4u /* the caller expects no bin list longer than 1024 entries: the (normally idle) 1024-thread sort
                                    launch is skipped.  If a longer list does turn up, the scan reports it as an OVERFLOW (nothing is
                                    rendered; max_tile_count > 1024 in the header tells the two cases apart): call again without it */
#define GSR_FLAG_COMPOSITE_TILES 8u /* compositing kernels that take the exponents of all (pixel, splat) pairs from bf16 matrix-core
                                      tiles (gsr_composite_tiles.hip, exact split evaluation) instead of computing them per pair on the
                                      vector ALUs (gsr_composite.hip).  Forward and backward of one view must agree on it. */
#define GSR_FLAG_NO_COLOR_GRAD 256u /* gsr_backward: the caller does not need dL_dcolors (stage 2: the colours are input pixels, which
                                       train_stage2.py never differentiates; torch: colors_precomp.requires_grad is False).  The tile family then
                                       leaves the three colour sums per (pixel, splat) out; dL_dcolors is written as zeros, every other gradient is
                                       bit-identical.  The VALU family ignores the flag (computes everything). */
#define GSR_FLAG_WAVE_PRIORITY 512u /* tile family: the compositing waves set their hardware priority (s_setprio) from the work they still
                                       have in front of them -- forward: more list left = higher priority, so that the waves resident on a SIMD
                                       finish together instead of one after the other (the arbiter serves the oldest wave first); backward:
                                       workgroups dispatched after the first resident generation overtake the leftovers of that generation.
                                       Results are unchanged.  It pays when a view's kernels have the chip to themselves (measured, config 2:
                                       forward 55 -> 51 us, backward 114 -> 110 us); with several views' kernels overlapped on other streams it
                                       costs ~2 % of the aggregate rate, so it is opt-in per call. */
#define GSR_FLAG_TIMING_STAGE(k) (GSR_FLAG_TIMING | (((unsigned)(k) + 1u) << 4)) /* ... or only stage k (GSR_STAGE_*) */

/* stage ids reported by gsr_timing_read() */
enum {
    GSR_STAGE_PREPROCESS = 0, GSR_STAGE_SCAN, GSR_STAGE_SCATTER, GSR_STAGE_SORT, GSR_STAGE_COMPOSITE_FWD,
    GSR_STAGE_COMPOSITE_BWD, GSR_STAGE_PREPROCESS_BWD, GSR_STAGE_COUNT
};

int gpsgs_abi_version(void);
/* Development aid: rows_device = 2 * bins rows of 4 u64 (zeroed by the caller), or NULL to switch it off.  While set, every tile-compositing
 * workgroup that did work records {wall clock at start, at end (100 MHz), shader cycles, HW_ID | XCC_ID << 32 | list length << 40} in row
 * blockIdx (forward) / bins + blockIdx (backward): the per-SIMD timeline tools/wg_trace.py analyses.  Process-wide, not thread-safe. */
int gsr_debug_set_wg_trace(unsigned long long *rows_device);

/* Diagnostic (synchronises `stream`): the shader clock in MHz under a chip-filling VALU load of a few milliseconds, from the ratio of
 * the per-cycle counter (s_memtime) to the constant-rate wall clock.  scratch3_device: 24 bytes of device memory.  bench.py prices the
 * VALU issue rate with it instead of assuming the 2.4 GHz maximum. */
int gpsgs_measure_sclk(unsigned long long *scratch3_device, double *mhz_host, void *stream);
const char *gpsgs_build_info(void); /* "gfx950 <compiler> <date>" */

/* ---- rasteriser ---------------------------------------------------------------------------------------------
 * Workspace: one caller-owned device buffer (>= gsr_workspace_bytes, 256-byte aligned) that carries the forward's
 * state to the backward (what upstream keeps in geomBuffer / binningBuffer / imgBuffer).
 * `instance_capacity` bounds R = number of (Gaussian, 8x8-pixel bin) instances the buffer can hold.  R is data dependent; the
 * forward never reads it back.  Instead the header records the R that was needed and an overflow flag:
 *   - overflow == 0: results are exact;
 *   - overflow != 0: NOTHING was rendered (out_color untouched apart from zero fill); the caller must re-run with
 *     instance_capacity >= max(num_rendered, num_slots) as reported.  (Upstream sizes the buffer with a blocking D2H read of R on every call.)
 */
typedef struct GsrHeader {      /* first bytes of the workspace, device memory */
    uint64_t num_rendered;      /* R needed by the last gsr_forward */
    uint32_t overflow;          /* 1 if R (or, with a backward tail, num_slots) > instance_capacity */
    uint32_t max_tile_count;    /* longest per-bin list (one 8x8-pixel bin = one wave64 work item) */
    uint32_t num_busy_wgs;      /* bins with a non-empty list; they are scheduled first */
    uint32_t num_slots;         /* training workspaces only: bin-rect cells of all Gaussians (one gradient-record slot each); also <= capacity */
    uint32_t num_points;        /* Gaussians of this view: P, or end - begin of GsrViewExt.row_range */
    uint32_t row_overflow;      /* 1 if a row range held more rows than the P (= row capacity) the call was made with: reported as overflow */
    uint32_t reserved[8];       /* scratch of the library (direct lists: the accumulators of the totals workgroups); zeroed by every forward */
} GsrHeader;

/* direct lists (GsrViewExt.bin_capacity): 1 if this image size and capacity can use them (capacity a multiple of 64 in 64..1024, at most 65,536 bins) */
int gsr_direct_lists_ok(int width, int height, uint32_t bin_capacity);
/* workspace size for either list form (bin_capacity 0 = scanned: then equal to the two functions below); 0 on invalid arguments.  With direct lists
 * instance_capacity still bounds the gradient-record slots of a training workspace (header.num_slots); the lists themselves take bins x bin_capacity entries */
size_t gsr_workspace_bytes_ex(int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity, int forward_only);
size_t gsr_workspace_bytes(int P, int width, int height, int64_t instance_capacity);              /* forward + backward */
size_t gsr_workspace_bytes_forward_only(int P, int width, int height, int64_t instance_capacity); /* inference: no backward tail; a forward on
                                                                                                     such a workspace also skips the per-pixel state
                                                                                                     only a backward reads (final T, last contributor) */

/* Forward.  Inputs fp32, contiguous: means3D[P,3], colors[P,3], opacities[P], scales[P,3], rotations[P,4] (w,x,y,z;
 * NOT re-normalised), viewmatrix[16], projmatrix[16] (flat column-major = the transposed tensors the reference
 * passes), bg[3].  Outputs: out_color[3,H,W], radii[P].  P == 0: out_color is zero-filled (not background). */
int gsr_forward(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream);

/* gsr_forward with EARLY capacity notification.  `host_header_out` is 32 bytes of PINNED host memory the device can write
 * (hipHostMalloc / hipHostRegister; torch pin_memory()).  As soon as the binning scan knows the instance count -- about a fifth
 * of the way into the forward, before scatter / sort / compositing run -- the device stores the first 28 header bytes
 * {u64 num_rendered; u32 overflow, max_tile_count, num_busy_wgs (0 with direct lists: the work order is made later), num_slots, 0} there and then release-stores `notify_seq`
 * (non-zero, chosen by the caller, different from the word's current value) into the u32 at byte 28.  The host spins on that
 * word instead of waiting for the whole forward: the capacity check of the reference's blocking `num_rendered` readback
 * (rasterizer_impl.cu forward, cudaMemcpy of the scan total) then costs no GPU idle time.  With P == 0 nothing is written.
 * GPSGS_E_INVALID if the pointer is not device-visible pinned memory or notify_seq == 0. */
int gsr_forward_notify(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                       const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                       const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                       void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream,
                       void *host_header_out, uint32_t notify_seq);

/* Optional extras of one view (gsr_forward_ex / gsr_backward_ex); zero-initialise, NULL = none.
 *   row_range   DEVICE pointer to two u32 {begin, end}: the Gaussians of this view are rows [begin, end) of the per-Gaussian input
 *               arrays AND of the per-Gaussian outputs (radii, the six gradient arrays) -- the arrays are batch-wide, as
 *               lib/GaussianRender.py:15-34 would leave them if it did not split per sample -- and `P` is only a CAPACITY: it sizes the
 *               workspace and the launches, the kernels read the real count from the device.  The host then never needs the number of
 *               valid pixels of a sample (the reference learns it through ten boolean-mask gathers = ten device syncs per sample).
 *               end - begin > P is reported like an instance overflow (header: row_overflow = 1, num_points = end - begin).
 *   order_hint  longest per-bin list of an earlier, similar view (header.max_tile_count), 0 = unknown.  The scan dispatches the
 *               bins whose list exceeds 1/2 and 1/4 of it first (a bin is one wave's sequential job: longest-processing-time-first).
 *               Decides the ORDER of the work only, never a result. */
typedef struct GsrViewExt {
    const uint32_t *row_range;
    uint32_t order_hint;
    /* ---- ABI 3: appearance / covariance inputs the reference never passes but its rasteriser module accepts ---------------------------
     *   shs            DEVICE [rows, sh_coeffs, 3] real spherical-harmonics coefficients (sh_coeffs <= 16), evaluated up to degree
     *                  sh_degree (0..3, (sh_degree + 1)^2 <= sh_coeffs) in the direction campos -> mean, + 0.5, clamped at 0 (upstream
     *                  computeColorFromSH).  With shs the `colors` argument of the call must be NULL (exactly one of the two), campos
     *                  (DEVICE [3], the settings' camera centre) is required, and the backward writes dL_dsh [rows, sh_coeffs, 3]
     *                  (zeros beyond the active degree and for invisible Gaussians) and adds the view-direction term to dL_dmeans3D;
     *                  its dL_dcolors argument may then be NULL.
     *   cov3D_precomp  DEVICE [rows, 6] upper triangles (xx, xy, xz, yy, yz, zz), used as given (scale_modifier does not apply).  With it
     *                  `scales` and `rotations` must be NULL (exactly one of the two forms); the backward writes dL_dcov3D [rows, 6]
     *                  (upstream's convention: an off-diagonal entry carries both symmetric positions) and leaves dL_dscales / dL_drotations
     *                  untouched (they may be NULL). */
    uint32_t sh_degree;
    uint32_t sh_coeffs;
    /* ---- ABI 4: DIRECT bin lists.  0 = scanned lists (every earlier version).  > 0 (a multiple of 64, <= 1024; gsr_direct_lists_ok()): every 8x8-pixel bin
     *   owns a fixed-capacity segment of bin_capacity list entries, so an instance's slot follows from the base the count atomic of the preprocess returned
     *   without any offsets: the scatter needs no scan in front of it, and what is left of the scan (header, work order, slot prefix) rides in the scatter
     *   launch; with the tile compositing family the forward compositing waves sort their own lists, so there is no sort launch either -- three launches
     *   in front of the compositing instead of five, none of them a latency chain.  The workspace must be sized with gsr_workspace_bytes_ex(..., bin_capacity, ...); forward and backward of a view must
     *   pass the same value.  A view whose longest list exceeds bin_capacity is NOT rendered: header.overflow = 1 with header.max_tile_count >
     *   bin_capacity -- repeat it with a larger capacity or with scanned lists (the capacity question upstream answers with a blocking read of R). */
    uint32_t bin_capacity;
    const float *shs;
    const float *campos;
    const float *cov3D_precomp;
    float *dL_dsh;      /* gsr_backward_ex only */
    float *dL_dcov3D;   /* gsr_backward_ex only */
    uint32_t reserved[4];
} GsrViewExt;

/* gsr_forward_notify + GsrViewExt (host_header_out may be NULL: no notification, like gsr_forward).  The early header's word 6 carries
 * num_points. */
int gsr_forward_ex(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                   const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                   const float *viewmatrix, const float *projmatrix, const float *bg, float *out_color, int *radii,
                   void *workspace, size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream,
                   void *host_header_out, uint32_t notify_seq, const GsrViewExt *ext);

/* Backward.  Same inputs and the workspace left by the matching gsr_forward, plus dL_dpix[3,H,W] (contiguous).
 * Writes (does not accumulate) dL_dmeans3D[P,3], dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P],
 * dL_dscales[P,3], dL_drotations[P,4]. */
int gsr_backward(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                 const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                 const float *viewmatrix, const float *projmatrix, const float *bg, const int *radii,
                 const float *dL_dpix, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors,
                 float *dL_dopacity, float *dL_dscales, float *dL_drotations, void *workspace,
                 size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream);

/* gsr_backward + GsrViewExt (the same row_range the forward was given). */
int gsr_backward_ex(int P, int width, int height, const float *means3D, const float *colors, const float *opacities,
                    const float *scales, const float *rotations, float scale_modifier, float tanfovx, float tanfovy,
                    const float *viewmatrix, const float *projmatrix, const float *bg, const int *radii,
                    const float *dL_dpix, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors,
                    float *dL_dopacity, float *dL_dscales, float *dL_drotations, void *workspace,
                    size_t workspace_bytes, int64_t instance_capacity, unsigned flags, void *stream, const GsrViewExt *ext);

/* Visibility mask (upstream `_C.mark_visible`, reached through GaussianRasterizer.markVisible(positions) of the module the reference imports at
 * gaussian_renderer/__init__.py:14; the reference itself never calls it): present[i] = 1 iff point i passes the near-plane test of the forward
 * (view-space z > 0.2), else 0.  means3D[P,3], viewmatrix[16] / projmatrix[16] as for gsr_forward (projmatrix is accepted for signature parity and
 * not read: upstream's in_frustum() tests the view-space depth only). */
int gsr_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present, void *stream);

/* Enqueues a copy of the first 32 header bytes {u64 num_rendered, u32 overflow, max_bin_count, num_busy, num_slots, 2 pad} to PINNED host memory on
 * `stream`; does not synchronise (the host reads it after an event / stream sync of its own). */
int gsr_copy_header_async(const void *workspace, void *host_pinned_out, void *stream);

/* Blocking helper for non-torch hosts: copies the header to host memory and synchronises `stream`. */
int gsr_read_header(const void *workspace, GsrHeader *host_out, void *stream);

/* Diagnostic: runs the matrix-core building blocks of the tile compositing kernels (coefficient split, operand arrangement, bf16
 * MFMA tiles, lane exchange) on pseudo-random splats against the quadratic form evaluated per lane in fp64.  out4 (device): {max
 * |error| / (1 + |value|) over 64 splats x 64 pixels, the same over pairs that can pass the alpha test, 1 if v_permlane32_swap behaves
 * as documented else 0, the largest |c0| met}. */
int gsr_selftest(float *out4_device, void *stream);

/* Diagnostic: after a gsr_backward on a training workspace, out2_device[0] = gradient-record slots that hold a record (one per (Gaussian, bin) instance
 * that received gradient), out2_device[1] = slots of the view (header.num_slots).  Enqueues a 16-byte memset + one kernel; does not synchronise. */
int gsr_debug_count_records(const void *workspace, size_t workspace_bytes, int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity,
                            unsigned long long *out2_device, void *stream); /* GPSGS_E_WORKSPACE for a forward-only workspace (it has no record slots) */

/* Profiling helper (not thread safe, not for use under graph capture): synchronises, then adds up the hipEvent
 * durations recorded by calls that carried GSR_FLAG_TIMING since the last read.  ms_sum[GSR_STAGE_COUNT] receives the
 * total milliseconds per stage, launches[GSR_STAGE_COUNT] the number of timed launches.  Resets the recorder. */
int gsr_timing_read(float *ms_sum_host, int *launches_host);

/* Debug/parity helper: copies selected intermediate arrays out of the workspace into caller DEVICE buffers (any may
 * be NULL): depth[P], xy[P,2], conic_opacity[P,4], rect[P,4] (int32 bx0,by0,bx1,by1: the 8x8-pixel BIN rect the Gaussian
 * is listed in), tile_ranges[NB,2] (int64 list range per bin; NB = (ceil(W/8) rounded up to 4) * ceil(H/8)),
 * point_list[instance_capacity, or NB * bin_capacity with direct lists: indexed by tile_ranges] (uint32, sorted per bin), final_T[H,W], n_contrib[H,W] (the last two are only produced by a
 * forward on a workspace with the backward tail). */
int gsr_export_state(const void *workspace, int P, int width, int height, int64_t instance_capacity, uint32_t bin_capacity, float *depth,
                     float *xy, float *conic_opacity, int *rect, int64_t *tile_ranges, uint32_t *point_list,
                     float *final_T, uint32_t *n_contrib, void *stream);

/* ---- fused mask-compaction + pack of the per-pixel Gaussian maps (lib/GaussianRender.py:15-34) -------------------------
 * For every batch element b and view v (lmain, rmain): the pixels with valid != 0, in raster order, become consecutive
 * rows; row order is sample, then view, then pixel -- exactly the order of the reference's mask-gathers + concats.
 * rgb rows are img * 0.5 + 0.5.  All inputs are described with ELEMENT strides so that permuted views (the reference's xyz
 * is one) need no copy: element (b, pixel, channel) = ptr[b*batch_stride + pixel*pixel_stride + channel*channel_stride].
 * Outputs: packed rows out_xyz[rows,3], out_rgb[rows,3], out_rot[rows,4], out_scale[rows,3], out_opacity[rows] with capacity
 * B*n_views*S2 rows; row_of_pixel[B,n_views,S2] (row index or ~0; kept for the backward); sample_offsets[B+1] (device):
 * sample b owns rows [sample_offsets[b], sample_offsets[b+1]).  No host synchronisation. */
typedef struct GsrStrided {
    const void *ptr;
    int64_t batch_stride, pixel_stride, channel_stride;
} GsrStrided;

size_t gsr_pack_scratch_bytes(int B, int n_views, int S2);
int gsr_pack_views(int B, int n_views, int S2, const GsrStrided *valid /*u8*/, const GsrStrided *xyz, const GsrStrided *img,
                   const GsrStrided *rot, const GsrStrided *scale, const GsrStrided *opacity /* host arrays [n_views] */,
                   float *out_xyz, float *out_rgb, float *out_rot, float *out_scale, float *out_opacity, uint32_t *row_of_pixel,
                   uint32_t *sample_offsets, uint32_t *scratch, void *stream);
/* Backward: packed row gradients (any may be NULL = zero) -> planar map gradients, contiguous: d_xyz[v] is [B,S2,3], the
 * others [B,C,S2]; host arrays of n_views device pointers, entries / arrays may be NULL (not wanted).  Invalid pixels get 0. */
int gsr_pack_views_backward(int B, int n_views, int S2, const uint32_t *row_of_pixel, const float *g_xyz, const float *g_rgb,
                            const float *g_rot, const float *g_scale, const float *g_opacity, float *const *d_xyz,
                            float *const *d_img, float *const *d_rot, float *const *d_scale, float *const *d_opacity, void *stream);

/* ---- fused L1 + SSIM loss (lib/loss.py:36-83 as used at train_stage2.py:70-72) ------------------------------------------
 * pred, gt: contiguous fp32 [planes = B*C, H, W].  out2 (device) = {mean |pred - gt|, mean SSIM map} (11x11 Gaussian window,
 * sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2).  m1,m2,m3 [planes,H,W]: derivative maps kept for the backward (all three
 * or none).  scratch: fl_scratch_bytes().  Backward: grad_out2 (device) = {dL/d(mean L1), dL/d(mean SSIM)}; writes d_pred. */
size_t fl_scratch_bytes(int planes, int H, int W);
int fl_l1_ssim_forward(const float *pred, const float *gt, int planes, int H, int W, float *m1, float *m2, float *m3, void *scratch,
                       float *out2, void *stream);
int fl_l1_ssim_backward(const float *pred, const float *gt, const float *m1, const float *m2, const float *m3, int planes, int H, int W,
                        const float *grad_out2, float *d_pred, void *stream);

/* ---- fused disparity -> inverse depth -> world points (lib/utils.py:88-120, lib/network.py:66-69) ---------------------
 * flow [B,1,S,S], mask channel 0 of [B,C,S,S] (mask_batch_stride = C*S*S elements), camera parameters on the HOST:
 * ref_intr/intr [B,3,3], extr [B,3,4] row-major, Tf_x [B] (any B: launched 16 samples at a time).  Outputs depth [B,1,S,S] (inverse depth), xyz [B,S*S,3],
 * valid [B,S*S] (u8).  Backward: g_depth [B,1,S,S] and/or g_xyz (element strides given; either may be NULL) -> d_flow. */
int up_unproject_forward(int B, int S, const float *flow, const float *mask, int64_t mask_batch_stride, const float *ref_intr_host,
                         const float *intr_host, const float *extr_host, const float *tf_host, float *depth, float *xyz,
                         uint8_t *valid, void *stream);
int up_unproject_backward(int B, int S, const float *depth, const float *mask, int64_t mask_batch_stride, const float *ref_intr_host,
                          const float *intr_host, const float *extr_host, const float *tf_host, const float *g_depth,
                          const float *g_xyz, int64_t gx_batch_stride, int64_t gx_pixel_stride, int64_t gx_channel_stride,
                          float *d_flow, void *stream);
/* The same with the cameras in DEVICE memory, cams_dev[B][31] = {ref_intr 3x3, intr 3x3, rows 0..2 of extr (3x4, row-major), Tf_x} per sample: nothing is
 * read on the host (the reference keeps these tensors on the GPU, train_stage2.py:154-156; reading them back would synchronise in the middle of the network
 * forward), one launch for any B, bit-identical results. */
int up_unproject_forward_dev(int B, int S, const float *flow, const float *mask, int64_t mask_batch_stride, const float *cams_dev, float *depth, float *xyz,
                             uint8_t *valid, void *stream);
int up_unproject_backward_dev(int B, int S, const float *depth, const float *mask, int64_t mask_batch_stride, const float *cams_dev, const float *g_depth,
                              const float *g_xyz, int64_t gx_batch_stride, int64_t gx_pixel_stride, int64_t gx_channel_stride, float *d_flow, void *stream);

/* ---- 1-D correlation sampler ----------------------------------------------------------------------------------
 * volume[N,H1,W1,W2], coords[N,H1,W1] fp32 (channel 0 of the reference's [N,1,H1,W1]), out[N,2r+1,H1,W1].
 * dtype: 0 = fp32, 1 = fp16 (volume / out / grads; coords always fp32). */
int cs_forward(const void *volume, const float *coords, void *out, int N, int H1, int W1, int W2, int radius,
               int dtype, void *stream);
int cs_backward(const float *coords, const void *grad_out, void *grad_volume, int N, int H1, int W1, int W2,
                int radius, int dtype, void *stream);

/* ---- correlation volume + pyramid, fused multi-level lookup, convex upsampling ----------
 * fmap1[N,D,H,W1], fmap2[N,D,H,W2] (contiguous NCHW, the reference's fmap12 / fmap21); pyramid[l] is [N,H,W1,W2>>l] for
 * l < levels (1..4), a HOST array of device pointers.  level 0 = sum_d fmap1 fmap2 / sqrt(D); level l+1 = average of
 * neighbouring pairs of level l along W2 (avg_pool2d([1,2], stride [1,2]): an odd last column is dropped).
 * dtype 0 = fp32, 1 = fp16 (inputs, pyramid and gradients; accumulation is always fp32, one rounding on store). */
int cv_build_forward(const void *fmap1, const void *fmap2, void *const *pyramid, int N, int D, int H, int W1, int W2, int levels,
                     int dtype, void *stream);
/* grad_pyramid[l] may be NULL (level received no gradient); grad_fmap1 / grad_fmap2 may be NULL (not needed); both are
 * written, not accumulated. */
int cv_build_backward(const void *fmap1, const void *fmap2, const void *const *grad_pyramid, void *grad_fmap1, void *grad_fmap2, int N,
                      int D, int H, int W1, int W2, int levels, int dtype, void *stream);
/* out[N, levels*(2r+1), H1, W1]: channel l*(2r+1)+k = tap k of level l sampled at coords / 2^l (cs_forward semantics per level). */
int cs_lookup_forward(const void *const *pyramid, const float *coords, void *out, int N, int H1, int W1, int W2, int levels, int radius,
                      int dtype, void *stream);
/* writes (zero-fills + taps) every grad_pyramid[l][N,H1,W1,W2>>l] */
int cs_lookup_backward(const float *coords, const void *grad_out, void *const *grad_pyramid, int N, int H1, int W1, int W2, int levels,
                       int radius, int dtype, void *stream);
/* flow[N,C,H,W] (C <= 4 forward, <= 2 backward: the reference upsamples a 2-channel flow), mask[N,9*f*f,H,W] logits,
 * out[N,C,f*H,f*W]; fp32.  The backward needs cu_upsample_scratch_bytes() of device scratch when grad_flow is requested
 * (per coarse cell and tap: the sums of softmax weight x upstream gradient); grad_flow / grad_mask may each be NULL. */
int cu_upsample_forward(const float *flow, const float *mask, float *out, int N, int C, int H, int W, int factor, void *stream);
size_t cu_upsample_scratch_bytes(int N, int C, int H, int W);
int cu_upsample_backward(const float *flow, const float *mask, const float *grad_out, float *grad_flow, float *grad_mask, void *scratch,
                         int N, int C, int H, int W, int factor, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GPSGS_H */
