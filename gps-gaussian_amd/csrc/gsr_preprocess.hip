// gsr_preprocess.hip -- per-Gaussian stages of the rasteriser for gfx950 (HBM-bound maps, one thread per Gaussian).
//
//   k_preprocess      : cull, project, 3D covariance -> EWA 2D covariance -> conic, radius, tile rect; writes the
//                       48-byte splat record + radii and counts the Gaussian into every tile of its rect.
//                       Replaces upstream preprocessCUDA (+ the tiles_touched half of the binning); semantics:
//                       SURVEY.md section 9.1; call-site configuration /root/reference/gaussian_renderer/__init__.py:36-62.
//   k_preprocess_bwd  : conic/mean2D/colour partial sums -> dL/d(mean3D, scale, rotation, ...) (upstream
//                       computeCov2DCUDA + preprocessCUDA backward fused).  cov3D is recomputed
//                       from scale/rotation instead of being stored (saves 48 B/Gaussian of HBM round trip).
//
// This translation unit is compiled with -ffp-contract=off and correctly-rounded div/sqrt: every discrete decision
// (near cull, radius, tile rect, depth bits used for ordering) is then bit-identical to the fp32 CPU oracle.
#include "gsr_common.h"

#pragma clang fp contract(off)

namespace {

struct Cam {
    float v[16];
    float p[16];
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ view, const float *__restrict__ proj) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 16; i++) { c.v[i] = view[i]; c.p[i] = proj[i]; }  // wave-uniform -> scalar loads
    return c;
}

__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float R[3][3]) {
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag(s)^2 R^T as M^T M with M[k][i] = s_k R[i][k]
__device__ __forceinline__ void cov3d(const float s[3], const float R[3][3], float c6[6]) {
    float M[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) M[k][i] = s[k] * R[i][k];
#define GSR_SIG(i, j) (M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j])
    c6[0] = GSR_SIG(0, 0); c6[1] = GSR_SIG(0, 1); c6[2] = GSR_SIG(0, 2);
    c6[3] = GSR_SIG(1, 1); c6[4] = GSR_SIG(1, 2); c6[5] = GSR_SIG(2, 2);
#undef GSR_SIG
}

struct Ewa {
    float t[3], xmul, ymul, T[2][3];
};

__device__ __forceinline__ Ewa ewa_setup(const float pv[3], const float *v, float fx, float fy, float tanx, float tany) {
    Ewa e;
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    float cx = txtz < -limx ? -limx : txtz; cx = cx > limx ? limx : cx;
    float cy = tytz < -limy ? -limy : tytz; cy = cy > limy ? limy : cy;
    e.t[0] = cx * pv[2];
    e.t[1] = cy * pv[2];
    e.t[2] = pv[2];
    const float J00 = fx / e.t[2];
    const float J02 = -(fx * e.t[0]) / (e.t[2] * e.t[2]);
    const float J11 = fy / e.t[2];
    const float J12 = -(fy * e.t[1]) / (e.t[2] * e.t[2]);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        e.T[0][j] = J00 * v[j * 4 + 0] + J02 * v[j * 4 + 2];
        e.T[1][j] = J11 * v[j * 4 + 1] + J12 * v[j * 4 + 2];
    }
    return e;
}

__device__ __forceinline__ void cov2d(const Ewa &e, const float c6[6], float &a, float &b, float &c) {
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float V[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int m = 0; m < 2; m++) V[k][m] = S[k][0] * e.T[m][0] + S[k][1] * e.T[m][1] + S[k][2] * e.T[m][2];
    a = e.T[0][0] * V[0][0] + e.T[0][1] * V[1][0] + e.T[0][2] * V[2][0] + 0.3f;
    b = e.T[0][0] * V[0][1] + e.T[0][1] * V[1][1] + e.T[0][2] * V[2][1];
    c = e.T[1][0] * V[0][1] + e.T[1][1] * V[1][1] + e.T[1][2] * V[2][1] + 0.3f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }


// ---- spherical-harmonics colours (the `shs` input of the upstream interface; the reference itself passes colors_precomp,
// /root/reference/gaussian_renderer/__init__.py:54-62, and constructs the settings with sh_degree = 3 and campos at :46-47) -----------------
// colour = max(sum_k basis_k(dir) sh[k] + 0.5, 0), dir = (mean - campos) normalised; basis = the published real SH basis of the 3D-Gaussian-
// splatting rasteriser up to degree 3 (the CPU checker restates the same; the tests pin both against torch autograd).
constexpr float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f, SH_C3_6 = -0.5900435899266435f;

__device__ __forceinline__ int sh_basis(int deg, float x, float y, float z, float b[16]) {
    b[0] = SH_C0;
    if (deg < 1) return 1;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy); b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
    if (deg < 3) return 9;
    b[9] = SH_C3_0 * y * (3.f * xx - yy);
    b[10] = SH_C3_1 * xy * z;
    b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
    b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
    b[14] = SH_C3_5 * z * (xx - yy);
    b[15] = SH_C3_6 * x * (xx - 3.f * yy);
    return 16;
}
// d basis_k / d(x, y, z), the direction's components taken as free variables (the normalisation follows in the caller)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float d[16][3]) {
#pragma unroll
    for (int k = 0; k < 16; k++) { d[k][0] = 0.f; d[k][1] = 0.f; d[k][2] = 0.f; }
    if (deg < 1) return;
    d[1][1] = -SH_C1; d[2][2] = SH_C1; d[3][0] = -SH_C1;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    d[4][0] = SH_C2_0 * y; d[4][1] = SH_C2_0 * x;
    d[5][1] = SH_C2_1 * z; d[5][2] = SH_C2_1 * y;
    d[6][0] = SH_C2_2 * -2.f * x; d[6][1] = SH_C2_2 * -2.f * y; d[6][2] = SH_C2_2 * 4.f * z;
    d[7][0] = SH_C2_3 * z; d[7][2] = SH_C2_3 * x;
    d[8][0] = SH_C2_4 * 2.f * x; d[8][1] = SH_C2_4 * -2.f * y;
    if (deg < 3) return;
    d[9][0] = SH_C3_0 * 6.f * xy; d[9][1] = SH_C3_0 * 3.f * (xx - yy);
    d[10][0] = SH_C3_1 * yz; d[10][1] = SH_C3_1 * xz; d[10][2] = SH_C3_1 * xy;
    d[11][0] = SH_C3_2 * -2.f * xy; d[11][1] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); d[11][2] = SH_C3_2 * 8.f * yz;
    d[12][0] = SH_C3_3 * -6.f * xz; d[12][1] = SH_C3_3 * -6.f * yz; d[12][2] = SH_C3_3 * 3.f * (2.f * zz - xx - yy);
    d[13][0] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); d[13][1] = SH_C3_4 * -2.f * xy; d[13][2] = SH_C3_4 * 8.f * xz;
    d[14][0] = SH_C3_5 * 2.f * xz; d[14][1] = SH_C3_5 * -2.f * yz; d[14][2] = SH_C3_5 * (xx - yy);
    d[15][0] = SH_C3_6 * 3.f * (xx - yy); d[15][1] = SH_C3_6 * -6.f * xy;
}
// colour of one Gaussian from its coefficients sh[M][3] (row r of the batch-wide array); clamp[ch] = the sum was negative (gradient cut).
// Evaluated by the forward AND re-evaluated by the backward (same code, same uncontracted arithmetic, same translation unit: identical clamp
// decisions) instead of keeping 3 flags per Gaussian in the workspace.
__device__ __forceinline__ void sh_color(int deg, const float *__restrict__ sh, const float p[3], const float *__restrict__ campos, float rgb[3],
                                         bool clamp[3], float dir[3], float raw[3]) {
    raw[0] = p[0] - campos[0]; raw[1] = p[1] - campos[1]; raw[2] = p[2] - campos[2];
    const float len = sqrtf(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2]);
    dir[0] = raw[0] / len; dir[1] = raw[1] / len; dir[2] = raw[2] / len;
    float b[16];
    const int n = sh_basis(deg, dir[0], dir[1], dir[2], b);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++)  // (unrolled with a guard: the basis stays in registers)
            if (k < n) v += b[k] * sh[3 * k + ch];
        v += 0.5f;
        clamp[ch] = v < 0.f;
        rgb[ch] = v < 0.f ? 0.f : v;
    }
}

// The (Gaussian, bin) predicate of the COUNT pass: small rects (<= 32 cells) are tested per cell and the outcomes kept as a bit mask for k_scatter;
// large rects go by row intervals (gsr_row_cells) when the conic allows, per cell otherwise -- k_scatter re-derives either from the record
// and the threshold stored in the mask word (gsr_masked_hit).
struct CountHit {
    GsrHit h;
    GsrRowSpan rs;
    uint32_t *mask;
    int x0, y0, w;
    bool rows;
    __device__ __forceinline__ void span(int y, int &xa, int &xb) const {
        if (rows) gsr_row_cells(rs, y, xa, xb, xa, xb);
    }
    __device__ __forceinline__ bool operator()(int x, int y) const {
        if (rows) return true;
        const bool hh = gsr_bin_hit(h, x, y);
        const int k = (y - y0) * w + (x - x0);
        if (hh && k < 32) *mask |= 1u << k;
        return hh;
    }
};

// APPEAR = false: the inputs the reference passes (precomputed colours, scale + rotation): the instantiation every measured configuration runs.
// APPEAR = true: SH colours (q.shs) and / or precomputed 3D covariances (q.cov3D_precomp) -- its own instantiation, so that the common one carries
// neither the branches nor the registers of these inputs.
template <bool APPEAR>
__global__ __launch_bounds__(GSR_BIN_THREADS) void k_preprocess(GsrFwdParams q, GsrSplat *__restrict__ splats, uint4 *__restrict__ binrec,
                                                               uint32_t *__restrict__ wg_tab, uint32_t *__restrict__ bin_count, uint32_t *__restrict__ bin_count_fb,
                                                               GsrHeader *__restrict__ hdr) {
    const int i = blockIdx.x * GSR_BIN_THREADS + threadIdx.x;
    uint32_t rlo = 0, rhi = 0;
    float depth_out = 0.f;
    GsrHit hit = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f};
    // the view's Gaussians: rows [row0, row0 + nP) of the input / output arrays (row0 = 0, nP = P without a row range); everything
    // inside the workspace is indexed by i, the Gaussian's number inside the view
    uint32_t row0;
    int nP;
    gsr_view_rows(q.row_range, q.P, row0, nP);
    if (i == 0) {
        const uint32_t m = q.row_range ? q.row_range[1] - q.row_range[0] : (uint32_t)q.P;
        hdr->num_points = m;
        if (m > (uint32_t)q.P) hdr->row_overflow = 1u;  // more rows than the capacity the call was sized for: k_scan reports an overflow
    }
    if ((int)(blockIdx.x * GSR_BIN_THREADS) >= nP && nP < q.P) {
        // a row-range view is launched for its CAPACITY: a workgroup entirely behind the view's last Gaussian only leaves the neutral
        // entries the later kernels expect (slot prefix 0, empty bin box, empty masks) -- no barriers, no binning pass
        if (q.goff) {
            if (i < q.P) q.goff[i] = 0u;
            if (threadIdx.x == 0) q.gpart[blockIdx.x] = 0u;
        }
        if (i < q.P) binrec[i] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x < 4) wg_tab[(size_t)blockIdx.x * GSR_WG_TAB_WORDS + threadIdx.x] = 0u;
        return;
    }
    if (i < nP) {
    const size_t r = (size_t)row0 + (size_t)i;
    const Cam cam = load_cam(q.view, q.proj);
    const float p[3] = {q.means3D[3 * r], q.means3D[3 * r + 1], q.means3D[3 * r + 2]};
    const bool use_sh = APPEAR && q.shs != nullptr, use_cov = APPEAR && q.cov3D_precomp != nullptr;  // wave-uniform
    float col[3] = {0.f, 0.f, 0.f};
    if (!use_sh) { col[0] = q.colors[3 * r]; col[1] = q.colors[3 * r + 1]; col[2] = q.colors[3 * r + 2]; }
    const float op = q.opacities[r];
    // rotation and scale are fetched up front, together with the other inputs (one memory round trip instead of a second one
    // behind the near-plane test; a culled Gaussian wastes 28 bytes)
    float4 rot = make_float4(1.f, 0.f, 0.f, 0.f);
    float s_raw[3] = {0.f, 0.f, 0.f};
    if (!use_cov) {
        rot = *reinterpret_cast<const float4 *>(q.rotations + 4 * r);
        s_raw[0] = q.scales[3 * r]; s_raw[1] = q.scales[3 * r + 1]; s_raw[2] = q.scales[3 * r + 2];
    }
    // keep the compiler from sinking these loads back behind the branch
    __asm__ volatile("" : "+v"(rot.x), "+v"(rot.y), "+v"(rot.z), "+v"(rot.w), "+v"(s_raw[0]), "+v"(s_raw[1]), "+v"(s_raw[2]));
    const float sc[3] = {q.scale_modifier * s_raw[0], q.scale_modifier * s_raw[1], q.scale_modifier * s_raw[2]};

    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = make_float4(0.f, op, col[0], col[1]);
    float o2x = col[2], o2y = 0.f;
    int radius = 0;

    float pv[3];
    pv[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
    pv[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
    pv[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
    if (pv[2] > 0.2f) {  // near cull: the only frustum test upstream applies
        const float phx = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
        const float phy = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
        const float phw = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
        const float pw = 1.f / (phw + 0.0000001f);
        const float ppx = phx * pw, ppy = phy * pw;
        const float fx = q.fx, fy = q.fy;

        float R[3][3], c6[6];
        if (use_cov) {  // the covariance is an input (upper triangle xx, xy, xz, yy, yz, zz), used as given: scale_modifier does not apply
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = q.cov3D_precomp[6 * r + k];
        } else {
            quat_to_R(rot.x, rot.y, rot.z, rot.w, R);
            cov3d(sc, R, c6);
        }
        const Ewa e = ewa_setup(pv, cam.v, fx, fy, q.tanfovx, q.tanfovy);
        float a, b, c;
        cov2d(e, c6, a, b, c);
        const float det = a * c - b * b;
        if (det != 0.f) {
            const float det_inv = 1.f / det;
            const float mid = 0.5f * (a + c);
            float disc = mid * mid - det;
            if (disc < 0.1f) disc = 0.1f;
            const float l1 = mid + sqrtf(disc), l2 = mid - sqrtf(disc);
            const float my_radius = ceilf(3.f * sqrtf(l1 > l2 ? l1 : l2));
            const float px = ((ppx + 1.f) * (float)q.W - 1.f) * 0.5f;
            const float py = ((ppy + 1.f) * (float)q.H - 1.f) * 0.5f;
            const int r0x = clampi((int)((px - my_radius) / 16.f), 0, q.gx);
            const int r0y = clampi((int)((py - my_radius) / 16.f), 0, q.gy);
            const int r1x = clampi((int)((px + my_radius + 15.f) / 16.f), 0, q.gx);
            const int r1y = clampi((int)((py + my_radius + 15.f) / 16.f), 0, q.gy);
            if ((r1x - r0x) * (r1y - r0y) != 0) {
                radius = (int)my_radius;
                o0 = make_float4(px, py, c * det_inv, -b * det_inv);
                o1.x = a * det_inv;
                o2y = pv[2];
                if (use_sh) {  // only a visible Gaussian gets a colour evaluated (upstream preprocessCUDA does the same)
                    bool cl[3];
                    float dir[3], raw[3], rgb[3];
                    sh_color((int)q.sh_degree, q.shs + (size_t)3 * q.sh_coeffs * r, p, q.campos, rgb, cl, dir, raw);
                    o1.z = rgb[0]; o1.w = rgb[1]; o2x = rgb[2];
                }
                // Bin rect = (upstream's 16x16-tile rect, in 8-px bins) INTERSECT (bounding box of the alpha >= 1/255
                // level set).  alpha = op * exp(power) >= 1/255  <=>  d^T Sigma^-1 d <= 2 ln(255 op) =: 2 tau, whose
                // axis-aligned extent is |dx| <= sqrt(2 tau a), |dy| <= sqrt(2 tau c) with (a,b,c) the dilated 2D
                // covariance.  Pairs outside can never pass the alpha test, so not listing them is exact; the box is
                // inflated a little so that fp32 rounding in the compositing kernels can never disagree with it.
                // (v_log_f32 / v_sqrt_f32, 1 ulp each, instead of the library's logf and the correctly rounded square roots: ~70 vector instructions of a
                //  VALU-bound kernel, for numbers that are inflated by 0.1 - 0.2 % before anything is decided with them.  log2(1) is exactly 0 and the
                //  instruction is monotonic to its ulp, so the sign of tau -- opacity >= 1/255 -- is the exact one.)
                const float tau = __builtin_amdgcn_logf(255.f * op) * 0.693147180559945f;
                if (tau >= 0.f) {
                    const float hx = __builtin_amdgcn_sqrtf(2.f * tau * a) * 1.001f + 0.01f, hy = __builtin_amdgcn_sqrtf(2.f * tau * c) * 1.001f + 0.01f;
                    int b0x = (int)floorf((px - hx) / 8.f), b1x = (int)floorf((px + hx) / 8.f) + 1;
                    int b0y = (int)floorf((py - hy) / 8.f), b1y = (int)floorf((py + hy) / 8.f) + 1;
                    b0x = max(b0x, 2 * r0x); b1x = min(b1x, min(2 * r1x, q.bx_real));
                    b0y = max(b0y, 2 * r0y); b1y = min(b1y, min(2 * r1y, q.by));
                    if (b1x > b0x && b1y > b0y) {
                        rlo = (uint32_t)b0x | ((uint32_t)b0y << 16);
                        rhi = (uint32_t)b1x | ((uint32_t)b1y << 16);
                        hit = gsr_hit_setup(px, py, c * det_inv, -b * det_inv, a * det_inv, tau);  // from the STORED record values
                    }
                }
            }
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(splats + i);
    dst[0] = o0;
    dst[1] = o1;
    dst[2] = make_float4(o2x, o2y, __uint_as_float(rlo), __uint_as_float(rhi));
    depth_out = o2y;
    q.radii[r] = radius;
    }
    if (q.goff) {  // training workspace: the slot prefix the backward needs (gradient-record slots = bin-rect cells) falls out here
        __shared__ uint32_t s_w[GSR_BIN_THREADS / 64];
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        const int w = (int)(rhi & 0xffff) - (int)(rlo & 0xffff), h = (int)(rhi >> 16) - (int)(rlo >> 16);
        const uint32_t area = (w > 0 && h > 0) ? (uint32_t)(w * h) : 0u;
        uint32_t x = area;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wid] = x;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < GSR_BIN_THREADS / 64; k++) {
            const uint32_t v = s_w[k];
            if (k < wid) woff += v;
            tot += v;
        }
        if (i < q.P) q.goff[i] = woff + x - area;  // prefix inside this block of GSR_BIN_THREADS Gaussians
        if (tid == 0) q.gpart[blockIdx.x] = tot;   // k_scan_b / the scan waves of the direct lists' scatter launch turn these into the prefix of the blocks
    }
    // the exact ellipse/bin test of every cell of the rect is evaluated ONCE, here; the outcomes are kept as a bit mask
    // (cell k = row-major index inside the rect) that k_scatter reuses instead of re-testing every cell twice
    uint32_t mask = 0u;
    CountHit ch;
    ch.h = hit; ch.mask = &mask;
    ch.x0 = rlo & 0xffff; ch.y0 = rlo >> 16; ch.w = (int)(rhi & 0xffff) - ch.x0;
    const bool big_rect = ch.w * ((int)(rhi >> 16) - ch.y0) > 32;
    ch.rs.ok = 0;
    if (big_rect) ch.rs = gsr_rows_setup(hit);  // (a division and a square root nobody needs for the ~4-cell rects of trained scales)
    ch.rows = ch.rs.ok && big_rect;  // the rule of gsr_masked_hit(): k_scatter decides with the same numbers
    {
        __shared__ uint32_t s_cnt[GSR_BLOCK_TAB];
        __shared__ int s_box[4];
        gsr_block_bin<false, GSR_BIN_THREADS>(
            s_cnt, nullptr, s_box, rlo, rhi, q.bx, ch,
            // the RETURNED value is this workgroup's base inside the bin's list: recorded in wg_tab for k_scatter (which then needs no atomic of its own)
            [&](int bin, uint32_t cnt) { return atomicAdd(&bin_count[(size_t)bin * GSR_CPAD], cnt); },
            // a workgroup whose bins do not fit the table counts into the second array: its instances are placed behind the recorded ones
            [&](int bin, uint32_t cnt) { atomicAdd(&bin_count_fb[(size_t)bin * GSR_CPAD], cnt); return 0u; }, [](uint32_t, uint32_t, uint32_t) {},
            wg_tab + (size_t)blockIdx.x * GSR_WG_TAB_WORDS);
    }
    // rects of more than 32 cells are not cached: k_scatter re-tests their cells, with THIS threshold (see gsr_hit_from_threshold)
    if ((int)((rhi & 0xffff) - (rlo & 0xffff)) * (int)((rhi >> 16) - (rlo >> 16)) > 32) mask = __float_as_uint(hit.thr);
    // everything k_scatter needs of a Gaussian in ONE aligned 16-byte record {depth bits, bin rect, mask}: it used to gather 16 of the 48 bytes of
    // the splat record (a strided read that pulled most of the 29 MB array) + the mask word
    if (i < q.P) binrec[i] = make_uint4(__float_as_uint(depth_out), rlo, rhi, mask);
}

// APPEAR as k_preprocess: true = SH colours and / or precomputed covariances among the inputs.  DOPREC: the records were written without colour sums
// (GsrBwdParams::dop_in_record): dL/dopacity is their first float and inst_dop is not read -- a template parameter, not a run-time test: with the test
// inside the four-record gather the common instantiation went 28.7 -> 32.6 us (the conditional loads split the batch of twelve into two waits).
template <bool APPEAR, bool DOPREC>
__global__ __launch_bounds__(256) void k_preprocess_bwd(GsrBwdParams q, const GsrSplat *__restrict__ splats,
                                                        const uint32_t *__restrict__ goff, const uint32_t *__restrict__ gpart,
                                                        const uint8_t *__restrict__ inst_valid, const float *__restrict__ inst_dop,
                                                        const GsrGradAcc *__restrict__ inst_grad, const GsrHeader *__restrict__ hdr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t row0;
    int nP;
    gsr_view_rows(q.row_range, q.P, row0, nP);
    if (i >= nP) return;
    const size_t r = (size_t)row0 + (size_t)i;  // row of the batch-wide input / gradient arrays (= i without a row range)
    // an overflowed forward rendered nothing: inst_valid / inst_grad were never written (and the slot range may not even fit the
    // workspace), so every Gaussian gets an exact zero gradient instead of a gather over garbage
    const bool rendered = hdr->overflow == 0u;
    float dm[3] = {0.f, 0.f, 0.f}, dsc[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dcol[3] = {0.f, 0.f, 0.f}, dm2[2] = {0.f, 0.f}, dop = 0.f;
    const bool use_sh = APPEAR && q.shs != nullptr, use_cov = APPEAR && q.cov3D_precomp != nullptr;  // wave-uniform
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool sh_written = false;
    if (rendered && q.radii[r] > 0) {
        const Cam cam = load_cam(q.view, q.proj);
        // the per-Gaussian inputs of the chain rule are requested BEFORE the record gather, so they travel alongside it
        float in_p[3] = {q.means3D[3 * r], q.means3D[3 * r + 1], q.means3D[3 * r + 2]};
        float4 in_rot = make_float4(1.f, 0.f, 0.f, 0.f);
        float in_s[3] = {0.f, 0.f, 0.f};
        if (!use_cov) {
            in_rot = *reinterpret_cast<const float4 *>(q.rotations + 4 * r);
            in_s[0] = q.scales[3 * r]; in_s[1] = q.scales[3 * r + 1]; in_s[2] = q.scales[3 * r + 2];
        }
        __asm__ volatile("" : "+v"(in_p[0]), "+v"(in_p[1]), "+v"(in_p[2]), "+v"(in_rot.x), "+v"(in_rot.y), "+v"(in_rot.z), "+v"(in_rot.w),
                         "+v"(in_s[0]), "+v"(in_s[1]), "+v"(in_s[2]));
        // gather this Gaussian's instance records in rect order: fixed summation order -> reproducible gradients
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        float g2x = 0.f;
        // this Gaussian's slots [s0, s1): from the slot prefix alone (the splat record would cost a 48-byte-stride read for 8 bytes)
        const int gb = i >> GSR_BIN_SHIFT;
        const uint32_t gbase = gpart[gb];
        const uint32_t s0 = gbase + goff[i];
        const uint32_t s1 = ((i & (GSR_BIN_THREADS - 1)) != GSR_BIN_THREADS - 1 && i + 1 < q.P) ? gbase + goff[i + 1] : ((gb + 1) * GSR_BIN_THREADS < q.P ? gpart[gb + 1] : hdr->num_slots);
        // FLAGS FIRST, FOUR RECORDS PER STEP: the flags of 32 slots (two aligned 16-byte words -- the section is 256-byte aligned and padded, whatever
        // lies outside [s0, s1) is masked off -- both requested up front, the next pair while the current one is worked on), then the records of up to
        // four FLAGGED slots at a time, all twelve loads in flight, in slot order: a fixed summation order.  A slot nobody wrote costs its flag byte,
        // not 37 bytes: at config 2 a third of the slots (the splat is hidden, or below 1/255, in that bin), in the untrained-heads regime three
        // quarters.  Until round 5 short runs were read unconditionally (4 slots per step, what does not exist dropped by a select) and only long runs
        // flags-first, ONE record per step: this form measured 37.1 -> 32.1 us at config 2, 69 -> 60 rendered at 2048^2, 217 -> 191 at config 5,
        // 305 -> 278 in the regime (tools/stage_times.py).  (Tried and reverted in round 4: streaming the flags through an LDS tile.)
        if (s1 > s0) {
            auto nib = [](uint32_t w) { w &= 0x01010101u; return (w | (w >> 7) | (w >> 14) | (w >> 21)) & 0xFu; };
            auto bits16 = [&](const uint4 &f) { return nib(f.x) | (nib(f.y) << 4) | (nib(f.z) << 8) | (nib(f.w) << 12); };
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
            uint32_t base = s0 & ~15u;
            uint4 fa = *reinterpret_cast<const uint4 *>(inst_valid + base);
            uint4 fb = base + 16u < s1 ? *reinterpret_cast<const uint4 *>(inst_valid + base + 16u) : zero4;
            for (; base < s1; base += 32u) {
                uint32_t m = bits16(fa) | (bits16(fb) << 16);
                const uint32_t lo = s0 > base ? s0 - base : 0u, hi = s1 - base;
                m &= ~((1u << lo) - 1u);
                if (hi < 32u) m &= (1u << hi) - 1u;
                const uint32_t nb = base + 32u;
                if (nb < s1) {
                    fa = *reinterpret_cast<const uint4 *>(inst_valid + nb);
                    fb = nb + 16u < s1 ? *reinterpret_cast<const uint4 *>(inst_valid + nb + 16u) : zero4;
                }
                while (m) {
                    uint32_t ri[4];
                    bool ok[4];
                    const uint32_t first = base + (uint32_t)__builtin_ctz(m);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        ok[u] = m != 0u;
                        ri[u] = ok[u] ? base + (uint32_t)__builtin_ctz(m) : first;  // (a missing one re-reads the first: a cache hit, dropped by the select)
                        m &= m - 1u;  // (0 stays 0)
                    }
                    float4 a0[4], a1[4];
                    float a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float4 *r = reinterpret_cast<const float4 *>(inst_grad + ri[u]);
                        a0[u] = r[0];
                        a1[u] = r[1];
                        if (!DOPREC) a2[u] = inst_dop[ri[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        g0.x += ok[u] ? a0[u].x : 0.f; g0.y += ok[u] ? a0[u].y : 0.f; g0.z += ok[u] ? a0[u].z : 0.f; g0.w += ok[u] ? a0[u].w : 0.f;
                        g1.x += ok[u] ? a1[u].x : 0.f; g1.y += ok[u] ? a1[u].y : 0.f; g1.z += ok[u] ? a1[u].z : 0.f; g1.w += ok[u] ? a1[u].w : 0.f;
                        g2x += ok[u] ? a2[u] : 0.f;
                    }
                }
            }
        }
        if (DOPREC) {  // records without colour sums: their first float is dL/dopacity (summed in the same slot order), the other two are zeros
            g2x = g0.x;
            g0.x = 0.f;
        }
        const float4 g2 = make_float4(g2x, 0.f, 0.f, 0.f);
        dcol[0] = g0.x; dcol[1] = g0.y; dcol[2] = g0.z;
        dm2[0] = g0.w; dm2[1] = g1.x;
        const float dxx = g1.y, dxy = g1.z, dyy = g1.w;
        dop = g2.x;

        const float p[3] = {in_p[0], in_p[1], in_p[2]};
        const float4 rot = in_rot;
        const float sv[3] = {q.scale_modifier * in_s[0], q.scale_modifier * in_s[1], q.scale_modifier * in_s[2]};
        float Rm[3][3], c6[6];
        if (use_cov) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = q.cov3D_precomp[6 * r + k];
#pragma unroll
            for (int a_ = 0; a_ < 3; a_++)
#pragma unroll
                for (int b_ = 0; b_ < 3; b_++) Rm[a_][b_] = 0.f;
        } else {
            quat_to_R(rot.x, rot.y, rot.z, rot.w, Rm);
            cov3d(sv, Rm, c6);
        }
        float pv[3];
        pv[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        pv[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        pv[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        const float fx = q.fx, fy = q.fy;
        const Ewa e = ewa_setup(pv, cam.v, fx, fy, q.tanfovx, q.tanfovy);
        float a, b, c;
        cov2d(e, c6, a, b, c);

        // (Tried: FMA contraction + 1-ulp reciprocals for the gradient-only part below, 886 -> 765 VALU instructions.  The chain to
        //  dL/dmean3D cancels strongly when the FoV clamp is active, and the fp32 oracle evaluates it uncontracted in this order: one
        //  element of the clamp-active fuzz family moved to 1.17e-3 of the gradient scale, above the 1e-3 bar.  Kept exact.)
        // conic -> cov2D (a,b,c)
        const float denom = a * c - b * b;
        const float d2inv = 1.f / (denom * denom + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f, dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float(*T)[3] = e.T;
        if (d2inv != 0.f) {
            dL_da = d2inv * (-c * c * dxx + 2.f * b * c * dxy + (denom - a * c) * dyy);
            dL_dc = d2inv * (-a * a * dyy + 2.f * a * b * dxy + (denom - a * c) * dxx);
            dL_db = d2inv * 2.f * (b * c * dxx - (denom + 2.f * b * b) * dxy + a * b * dyy);
            dc6[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
            dc6[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
            dc6[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
            dc6[1] = 2.f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2.f * T[1][0] * T[1][1] * dL_dc;
            dc6[2] = 2.f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2.f * T[1][0] * T[1][2] * dL_dc;
            dc6[4] = 2.f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2.f * T[1][1] * T[1][2] * dL_dc;
        }
        // cov2D = T Sigma T^T -> dL/dT -> dL/dJ -> dL/dt -> mean3D (through Rw^T)
        const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dT[2][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float s0 = T[0][0] * S[j][0] + T[0][1] * S[j][1] + T[0][2] * S[j][2];
            const float s1 = T[1][0] * S[j][0] + T[1][1] * S[j][1] + T[1][2] * S[j][2];
            dT[0][j] = 2.f * s0 * dL_da + s1 * dL_db;
            dT[1][j] = 2.f * s1 * dL_dc + s0 * dL_db;
        }
        const float *v = cam.v;
        const float dJ00 = dT[0][0] * v[0] + dT[0][1] * v[4] + dT[0][2] * v[8];
        const float dJ02 = dT[0][0] * v[2] + dT[0][1] * v[6] + dT[0][2] * v[10];
        const float dJ11 = dT[1][0] * v[1] + dT[1][1] * v[5] + dT[1][2] * v[9];
        const float dJ12 = dT[1][0] * v[2] + dT[1][1] * v[6] + dT[1][2] * v[10];
        const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = e.xmul * -fx * tz2 * dJ02;
        const float dty = e.ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.t[0]) * tz3 * dJ02 + (2.f * fy * e.t[1]) * tz3 * dJ12;
        dm[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;
        dm[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
        dm[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;

        // mean2D (NDC-scaled) -> mean3D through p_hom / (w + 1e-7)
        const float *pr = cam.p;
        const float mhw = pr[3] * p[0] + pr[7] * p[1] + pr[11] * p[2] + pr[15];
        const float mw = 1.f / (mhw + 0.0000001f);
        const float mul1 = (pr[0] * p[0] + pr[4] * p[1] + pr[8] * p[2] + pr[12]) * mw * mw;
        const float mul2 = (pr[1] * p[0] + pr[5] * p[1] + pr[9] * p[2] + pr[13]) * mw * mw;
        dm[0] += (pr[0] * mw - pr[3] * mul1) * dm2[0] + (pr[1] * mw - pr[3] * mul2) * dm2[1];
        dm[1] += (pr[4] * mw - pr[7] * mul1) * dm2[0] + (pr[5] * mw - pr[7] * mul2) * dm2[1];
        dm[2] += (pr[8] * mw - pr[11] * mul1) * dm2[0] + (pr[9] * mw - pr[11] * mul2) * dm2[1];

        if (use_sh) {
            // colour -> SH coefficients, and -> mean3D through the view direction (upstream computeColorFromSH backward): the colour and its
            // clamp decisions are re-evaluated exactly as the forward did
            bool cl[3];
            float dir[3], raw[3], rgb[3], bs[16], dbs[16][3];
            const float *sh = q.shs + (size_t)3 * q.sh_coeffs * r;
            sh_color((int)q.sh_degree, sh, p, q.campos, rgb, cl, dir, raw);
            const int nb = sh_basis((int)q.sh_degree, dir[0], dir[1], dir[2], bs);
            sh_basis_grad((int)q.sh_degree, dir[0], dir[1], dir[2], dbs);
            const float dRGB[3] = {cl[0] ? 0.f : dcol[0], cl[1] ? 0.f : dcol[1], cl[2] ? 0.f : dcol[2]};
            float ddir[3] = {0.f, 0.f, 0.f};
            float *out = q.dL_dsh + (size_t)3 * q.sh_coeffs * r;
#pragma unroll
            for (int k = 0; k < 16; k++) {  // sh_coeffs <= 16 (checked at the C-ABI); unrolled with guards: bs / dbs stay in registers
                if (k < (int)q.sh_coeffs) {
                    const bool act = k < nb;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        out[3 * k + ch] = act ? bs[k] * dRGB[ch] : 0.f;  // coefficients beyond the active degree: exact zeros
                        if (act) {
                            const float shv = sh[3 * k + ch];
#pragma unroll
                            for (int j = 0; j < 3; j++) ddir[j] += dbs[k][j] * shv * dRGB[ch];
                        }
                    }
                }
            }
            sh_written = true;
            const float sum2 = raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2];
            const float invsum32 = 1.f / sqrtf(sum2 * sum2 * sum2);
            dm[0] += ((sum2 - raw[0] * raw[0]) * ddir[0] - raw[1] * raw[0] * ddir[1] - raw[2] * raw[0] * ddir[2]) * invsum32;
            dm[1] += (-raw[0] * raw[1] * ddir[0] + (sum2 - raw[1] * raw[1]) * ddir[1] - raw[2] * raw[1] * ddir[2]) * invsum32;
            dm[2] += (-raw[0] * raw[2] * ddir[0] - raw[1] * raw[2] * ddir[1] + (sum2 - raw[2] * raw[2]) * ddir[2]) * invsum32;
        }
        if (use_cov) {
#pragma unroll
            for (int k = 0; k < 6; k++) dcov[k] = dc6[k];  // upstream's dL_dcov3D: the off-diagonal entries carry both symmetric positions
        }

        // cov3D -> scale, rotation: Sigma = R D R^T, D = diag(s^2); dL/dR = 2 G R D; dL/ds_k = 2 s_k (R^T G R)_kk
        // (with a precomputed covariance Rm = 0 and sv = 0: the results below are zeros and are not stored)
        const float Gs[3][3] = {{dc6[0], 0.5f * dc6[1], 0.5f * dc6[2]},
                                {0.5f * dc6[1], dc6[3], 0.5f * dc6[4]},
                                {0.5f * dc6[2], 0.5f * dc6[4], dc6[5]}};
        float GR[3][3], dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) GR[r][k] = Gs[r][0] * Rm[0][k] + Gs[r][1] * Rm[1][k] + Gs[r][2] * Rm[2][k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float rtgr = Rm[0][k] * GR[0][k] + Rm[1][k] * GR[1][k] + Rm[2][k] * GR[2][k];
            dsc[k] = 2.f * sv[k] * rtgr * q.scale_modifier;
#pragma unroll
            for (int r = 0; r < 3; r++) dR[r][k] = 2.f * GR[r][k] * sv[k] * sv[k];
        }
        const float qr = rot.x, qx = rot.y, qy = rot.z, qz = rot.w;
        dq[0] = 2.f * (qz * (dR[1][0] - dR[0][1]) + qy * (dR[0][2] - dR[2][0]) + qx * (dR[2][1] - dR[1][2]));
        dq[1] = 2.f * (qy * (dR[0][1] + dR[1][0]) + qz * (dR[0][2] + dR[2][0]) + qr * (dR[2][1] - dR[1][2])) - 4.f * qx * (dR[1][1] + dR[2][2]);
        dq[2] = 2.f * (qx * (dR[0][1] + dR[1][0]) + qr * (dR[0][2] - dR[2][0]) + qz * (dR[1][2] + dR[2][1])) - 4.f * qy * (dR[0][0] + dR[2][2]);
        dq[3] = 2.f * (qr * (dR[1][0] - dR[0][1]) + qx * (dR[0][2] + dR[2][0]) + qy * (dR[1][2] + dR[2][1])) - 4.f * qz * (dR[0][0] + dR[1][1]);
    }
    const size_t i3 = 3 * r;
    q.dL_dmeans3D[i3] = dm[0]; q.dL_dmeans3D[i3 + 1] = dm[1]; q.dL_dmeans3D[i3 + 2] = dm[2];
    q.dL_dmeans2D[i3] = dm2[0]; q.dL_dmeans2D[i3 + 1] = dm2[1]; q.dL_dmeans2D[i3 + 2] = 0.f;
    if (!APPEAR || q.dL_dcolors) { q.dL_dcolors[i3] = dcol[0]; q.dL_dcolors[i3 + 1] = dcol[1]; q.dL_dcolors[i3 + 2] = dcol[2]; }
    q.dL_dopacity[r] = dop;
    if (!use_cov) {
        q.dL_dscales[i3] = dsc[0]; q.dL_dscales[i3 + 1] = dsc[1]; q.dL_dscales[i3 + 2] = dsc[2];
        *reinterpret_cast<float4 *>(q.dL_drotations + 4 * r) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) q.dL_dcov3D[6 * r + k] = dcov[k];
    }
    if (use_sh && !sh_written) {  // invisible (or an overflowed forward): exact zeros, like every other gradient of this Gaussian
        float *out = q.dL_dsh + (size_t)3 * q.sh_coeffs * r;
        for (int k = 0; k < 3 * (int)q.sh_coeffs; k++) out[k] = 0.f;
    }
}

// upstream checkFrustum / markVisible (GaussianRasterizer.markVisible of the module the reference imports at
// /root/reference/gaussian_renderer/__init__.py:14): a point is "visible" iff it passes the near-plane test of the preprocess, view-space
// z > 0.2 -- the same uncontracted expression k_preprocess evaluates, so the mask equals "k_preprocess did not cull it at the near plane" bit for bit
__global__ __launch_bounds__(256) void k_mark_visible(int P, const float *__restrict__ means3D, const float *__restrict__ view, uint8_t *__restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
    const float pz = view[2] * x + view[6] * y + view[10] * z + view[14];
    present[i] = pz > 0.2f ? 1 : 0;
}

}  // namespace

void gsr_launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

void gsr_launch_preprocess(const GsrFwdParams &p, GsrSplat *splats, uint4 *binrec, uint32_t *wg_tab, uint32_t *bin_count, uint32_t *bin_count_fb, GsrHeader *hdr,
                           hipStream_t s) {
    if (p.P <= 0) return;
    const dim3 grid((p.P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS), block(GSR_BIN_THREADS);
    GsrFwdParams q = p;
    q.fx = (float)q.W / (2.f * q.tanfovx);  // (the same correctly-rounded fp32 division the kernel used to evaluate per thread)
    q.fy = (float)q.H / (2.f * q.tanfovy);
    if (p.shs || p.cov3D_precomp) hipLaunchKernelGGL(k_preprocess<true>, grid, block, 0, s, q, splats, binrec, wg_tab, bin_count, bin_count_fb, hdr);
    else hipLaunchKernelGGL(k_preprocess<false>, grid, block, 0, s, q, splats, binrec, wg_tab, bin_count, bin_count_fb, hdr);
}

void gsr_launch_preprocess_bwd(const GsrBwdParams &p, const GsrSplat *splats, const uint32_t *goff, const uint32_t *gpart,
                               const uint8_t *inst_valid, const float *inst_dop, const GsrGradAcc *inst_grad, const GsrHeader *hdr,
                               hipStream_t s) {
    if (p.P <= 0) return;
    GsrBwdParams q = p;
    q.fx = (float)q.W / (2.f * q.tanfovx);
    q.fy = (float)q.H / (2.f * q.tanfovy);
    const dim3 grid((q.P + 255) / 256), block(256);
    const bool appear = q.shs || q.cov3D_precomp;
    if (appear && q.dop_in_record) hipLaunchKernelGGL((k_preprocess_bwd<true, true>), grid, block, 0, s, q, splats, goff, gpart, inst_valid, inst_dop, inst_grad, hdr);
    else if (appear) hipLaunchKernelGGL((k_preprocess_bwd<true, false>), grid, block, 0, s, q, splats, goff, gpart, inst_valid, inst_dop, inst_grad, hdr);
    else if (q.dop_in_record) hipLaunchKernelGGL((k_preprocess_bwd<false, true>), grid, block, 0, s, q, splats, goff, gpart, inst_valid, inst_dop, inst_grad, hdr);
    else hipLaunchKernelGGL((k_preprocess_bwd<false, false>), grid, block, 0, s, q, splats, goff, gpart, inst_valid, inst_dop, inst_grad, hdr);
}
