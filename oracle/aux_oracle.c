/*
 * aux_oracle.c -- CPU restatements of the two small kernels either side of the rasteriser.
 * TEST INFRASTRUCTURE ONLY (same rules as gsr_oracle.c).
 *
 * 1. corr_sampler forward/backward.  The CUDA extension (princeton-vl/RAFT-Stereo sampler/, unpinned,
 *    absent from /root/reference) is called at /root/reference/core/corr.py:22,28.  Its semantics are
 *    those of the in-repo pure-torch path CorrBlock1D.__call__ (/root/reference/core/corr.py:127-146
 *    -> bilinear_sampler, /root/reference/core/utils/utils.py:59-74: grid_sample, align_corners=True,
 *    zero padding, height-1 image).  PINNED: tests/golden/corr_sampler_*.npz are outputs of that
 *    reference code run here (tests/golden/make_golden.py).
 *
 * 2. z-buffer point splat restating the Taichi kernel render_respective_color
 *    (/root/reference/lib/TaichiRender.py:13-24): per point, truncate (x,y) to the pixel grid, clamp to
 *    the image, keep the point with the largest inverse depth.  The original's colour write is racy
 *    across points; this restatement is the sequential order (point 0..N-1, '>=' so later ties win).
 *    taichi is not installed, so it is unpinned and only serves as the "Taichi/CPU raster" timing row.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* volume [N][H1][W1][W2], coords [N][H1][W1] (channel 0 of the reference's [N,1,H1,W1]), out [N][2r+1][H1][W1] */
void cs_oracle_forward(const float *volume, const float *coords, float *out, int N, int H1, int W1, int W2, int r) {
    const int rd = 2 * r + 1;
    memset(out, 0, sizeof(float) * (size_t)N * rd * H1 * W1);
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < H1; y++)
            for (int x = 0; x < W1; x++) {
                const float x0 = coords[((size_t)n * H1 + y) * W1 + x];
                const float fl = floorf(x0);
                const float dx = x0 - fl;
                const float *v = volume + (((size_t)n * H1 + y) * W1 + x) * W2;
                for (int i = 0; i < rd + 1; i++) {
                    const int x1 = (int)fl - r + i;
                    if (x1 < 0 || x1 >= W2) continue;
                    const float s = v[x1];
                    if (i > 0) out[(((size_t)n * rd + (i - 1)) * H1 + y) * W1 + x] += s * dx;
                    if (i < rd) out[(((size_t)n * rd + i) * H1 + y) * W1 + x] += s * (1.0f - dx);
                }
            }
}

/* grad_out [N][2r+1][H1][W1] -> grad_volume [N][H1][W1][W2]; no gradient to coords (core/corr.py:29) */
void cs_oracle_backward(const float *coords, const float *grad_out, float *grad_volume, int N, int H1, int W1, int W2, int r) {
    const int rd = 2 * r + 1;
    memset(grad_volume, 0, sizeof(float) * (size_t)N * H1 * W1 * W2);
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < H1; y++)
            for (int x = 0; x < W1; x++) {
                const float x0 = coords[((size_t)n * H1 + y) * W1 + x];
                const float fl = floorf(x0);
                const float dx = x0 - fl;
                float *gv = grad_volume + (((size_t)n * H1 + y) * W1 + x) * W2;
                for (int i = 0; i < rd + 1; i++) {
                    const int x1 = (int)fl - r + i;
                    if (x1 < 0 || x1 >= W2) continue;
                    float g = 0.0f;
                    if (i > 0) g += grad_out[(((size_t)n * rd + (i - 1)) * H1 + y) * W1 + x] * dx;
                    if (i < rd) g += grad_out[(((size_t)n * rd + i) * H1 + y) * W1 + x] * (1.0f - dx);
                    gv[x1] += g;
                }
            }
}

/* pts [B][N][6] = (x_pix, y_pix, inv_depth, r, g, b); mask [B][N]; depth [B][res][res] (init 0);
 * color [B][3][res][res] (init -1).  Called once per source view, like TaichiRender.py:56-57. */
void zsplat_oracle(const float *pts, const float *mask, float *depth, float *color, int B, int N, int res) {
#pragma omp parallel for
    for (int b = 0; b < B; b++)
        for (int i = 0; i < N; i++) {
            if (mask[(size_t)b * N + i] < 0.5f) continue;
            const float *p = pts + ((size_t)b * N + i) * 6;
            int ix = (int)p[0], iy = (int)p[1];
            ix = ix < 0 ? 0 : (ix > res - 1 ? res - 1 : ix);
            iy = iy < 0 ? 0 : (iy > res - 1 ? res - 1 : iy);
            float *d = depth + ((size_t)b * res + iy) * res + ix;
            if (p[2] >= *d) {
                *d = p[2];
                for (int k = 0; k < 3; k++) color[(((size_t)b * 3 + k) * res + iy) * res + ix] = p[3 + k];
            }
        }
}
