// Torch-free host for the C-ABI (include/gpsgs.h): what a C/C++ integrator of the render path writes.
// Build: hipcc --offload-arch=gfx950 tests/capi_host/capi_host_smoke.cpp -Iinclude -Lgps-gaussian_amd/lib -lgpsgs_hip -o <out>
// Renders a few hundred Gaussians forward + backward, re-runs after a deliberate capacity overflow, checks basic invariants
// and prints "CAPI_HOST_OK".  (Numerical parity is the job of the Python GPU tests; this proves the boundary stands alone.)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gpsgs.h"

#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); return 1; } } while (0)

template <typename T> static T *dev(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main() {
    const int P = 400, W = 100, H = 70;
    const float fx = 90.f;
    std::vector<float> m(3 * P), c(3 * P), o(P), s(3 * P), r(4 * P);
    srand(7);
    auto U = []() { return (float)rand() / (float)RAND_MAX; };
    for (int i = 0; i < P; i++) {
        m[3 * i] = (U() - 0.5f) * 2.0f; m[3 * i + 1] = (U() - 0.5f) * 1.4f; m[3 * i + 2] = 1.5f + 2.f * U();
        for (int k = 0; k < 3; k++) { c[3 * i + k] = U(); s[3 * i + k] = 0.01f + 0.05f * U(); }
        o[i] = 0.2f + 0.7f * U();
        float q[4] = {U() - .5f, U() - .5f, U() - .5f, U() - .5f}, n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; k++) r[4 * i + k] = q[k] / n;
    }
    // identity view; projection of lib/graphics_utils.py:31-48 (transposed -> flat column-major), znear 0.01, zfar 100
    std::vector<float> view(16, 0.f), proj(16, 0.f), bg = {0.1f, 0.2f, 0.3f};
    for (int i = 0; i < 4; i++) view[i * 4 + i] = 1.f;
    const float cx = W / 2.f, cy = H / 2.f, zn = 0.01f, zf = 100.f;
    // P[0][0]=2fx/W, P[1][1]=2fy/H, P[0][2]=(2cx-W)/W, P[1][2]=(2cy-H)/H, P[3][2]=1, P[2][2]=zf/(zf-zn), P[2][3]=-zf zn/(zf-zn); flat[col*4+row]
    proj[0 * 4 + 0] = 2 * fx / W; proj[1 * 4 + 1] = 2 * fx / H; proj[2 * 4 + 0] = (2 * cx - W) / W; proj[2 * 4 + 1] = (2 * cy - H) / H;
    proj[2 * 4 + 3] = 1.f; proj[2 * 4 + 2] = zf / (zf - zn); proj[3 * 4 + 2] = -zf * zn / (zf - zn);
    float *dm = dev(m), *dc = dev(c), *dop = dev(o), *ds = dev(s), *dr = dev(r), *dv = dev(view), *dp = dev(proj), *dbg = dev(bg);
    float *color; int *radii;
    CK(hipMalloc(&color, sizeof(float) * 3 * W * H)); CK(hipMalloc(&radii, sizeof(int) * P));
    hipStream_t st; CK(hipStreamCreate(&st));

    GsrHeader h;
    void *ws = nullptr; size_t nbytes = 0; int64_t cap = 16;  // far too small on purpose
    for (int attempt = 0; attempt < 2; attempt++) {
        nbytes = gsr_workspace_bytes(P, W, H, cap);
        if (ws) (void)hipFree(ws);
        CK(hipMalloc(&ws, nbytes));
        int rc = gsr_forward(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color, radii, ws, nbytes, cap, 0, st);
        if (rc != GPSGS_OK) { printf("gsr_forward rc=%d\n", rc); return 1; }
        if (gsr_read_header(ws, &h, st) != GPSGS_OK) return 1;
        if (attempt == 0) {
            if (!h.overflow) { printf("expected an overflow with capacity 16\n"); return 1; }
            cap = (int64_t)(h.num_rendered > h.num_slots ? h.num_rendered : h.num_slots) + 8;  // the header says how much was needed
        }
    }
    if (h.overflow || h.num_rendered == 0) { printf("second attempt: overflow=%u R=%llu\n", h.overflow, (unsigned long long)h.num_rendered); return 1; }
    std::vector<float> img(3 * W * H);
    CK(hipMemcpy(img.data(), color, img.size() * 4, hipMemcpyDeviceToHost));
    double sum = 0; for (float v : img) { if (!std::isfinite(v)) { printf("non-finite pixel\n"); return 1; } sum += v; }
    if (!(sum > 0.1 * W * H)) { printf("image looks empty: %f\n", sum); return 1; }

    std::vector<float> gpix(3 * W * H, 1.0f);
    float *dgp = dev(gpix), *g3, *g2, *gc, *go, *gs, *gr;
    CK(hipMalloc(&g3, 12 * P)); CK(hipMalloc(&g2, 12 * P)); CK(hipMalloc(&gc, 12 * P)); CK(hipMalloc(&go, 4 * P)); CK(hipMalloc(&gs, 12 * P)); CK(hipMalloc(&gr, 16 * P));
    int rc = gsr_backward(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, radii, dgp, g3, g2, gc, go, gs, gr, ws, nbytes, cap, 0, st);
    if (rc != GPSGS_OK) { printf("gsr_backward rc=%d\n", rc); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<float> hgc(3 * P);
    CK(hipMemcpy(hgc.data(), gc, 12 * P, hipMemcpyDeviceToHost));
    // d(sum of pixels)/d colour = sum over pixels of alpha*T >= 0, and > 0 for every Gaussian that was drawn
    double gsum = 0; for (float v : hgc) { if (!(v >= 0.f) || !std::isfinite(v)) { printf("bad colour gradient %f\n", v); return 1; } gsum += v; }
    if (!(gsum > 0)) { printf("zero gradient\n"); return 1; }
    // early notification: the scan kernel stores the header into pinned host memory; the host spins on the sequence word
    {
        volatile uint32_t *pin = nullptr;
        CK(hipHostMalloc((void **)&pin, 32, hipHostMallocDefault));
        for (int k = 0; k < 8; k++) pin[k] = 0;
        rc = gsr_forward_notify(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color, radii, ws, nbytes, cap, 0, st,
                                (void *)pin, 77u);
        if (rc != GPSGS_OK) { printf("gsr_forward_notify rc=%d\n", rc); return 1; }
        long spins = 0;
        while (pin[7] != 77u) {
            if (++spins > 2000000000L) { printf("notification never arrived\n"); return 1; }
        }
        const uint64_t Rn = (uint64_t)pin[0] | ((uint64_t)pin[1] << 32);
        if (Rn != h.num_rendered || pin[2] != 0u || pin[3] != h.max_tile_count || pin[4] != h.num_busy_wgs || pin[5] != h.num_slots) {
            printf("notified header differs: R=%llu ovf=%u max=%u busy=%u slots=%u\n", (unsigned long long)Rn, pin[2], pin[3], pin[4], pin[5]);
            return 1;
        }
        CK(hipStreamSynchronize(st));
        int not_pinned[8];
        if (gsr_forward_notify(P, W, H, dm, dc, dop, ds, dr, 1.f, 1.f, 1.f, dv, dp, dbg, color, radii, ws, nbytes, cap, 0, st, not_pinned, 5u) !=
            GPSGS_E_INVALID) { printf("pageable notify target was accepted\n"); return 1; }
        if (gsr_forward_notify(P, W, H, dm, dc, dop, ds, dr, 1.f, 1.f, 1.f, dv, dp, dbg, color, radii, ws, nbytes, cap, 0, st, (void *)pin, 0u) !=
            GPSGS_E_INVALID) { printf("notify_seq 0 was accepted\n"); return 1; }
        CK(hipHostFree((void *)pin));
    }
    // ABI version 2: the same view as a ROW RANGE of batch-wide arrays (what a pts2render-style caller has after packing a batch): the
    // Gaussians sit in rows [P, 2P) of arrays of 3P rows whose other rows hold NaNs, {begin, end} lives in device memory, the call is sized
    // for a capacity of P + 100 rows.  Image and gradients must equal the plain call bit for bit; without the colour gradient
    // (GSR_FLAG_NO_COLOR_GRAD) every other gradient must still be identical and dL_dcolors must be zero.
    {
        if (gpsgs_abi_version() != 4) { printf("ABI version %d\n", gpsgs_abi_version()); return 1; }
        const float nanv = std::nanf("");
        auto wide = [&](const std::vector<float> &src, int ch) {
            std::vector<float> w((size_t)3 * P * ch, nanv);
            for (size_t k = 0; k < (size_t)P * ch; k++) w[(size_t)P * ch + k] = src[k];
            return w;
        };
        float *wm = dev(wide(m, 3)), *wc = dev(wide(c, 3)), *wo = dev(wide(o, 1)), *wsc = dev(wide(s, 3)), *wr = dev(wide(r, 4));
        std::vector<uint32_t> range = {(uint32_t)P, (uint32_t)(2 * P)};
        uint32_t *drange = dev(range);
        const int Pcap = P + 100;
        const size_t nb2 = gsr_workspace_bytes(Pcap, W, H, cap);
        void *ws2; float *color2; int *radii2;
        CK(hipMalloc(&ws2, nb2)); CK(hipMalloc(&color2, sizeof(float) * 3 * W * H)); CK(hipMalloc(&radii2, sizeof(int) * 3 * P));
        // the plain calls once more with the default kernel family (exponents from the matrix-core tiles): the comparison baseline
        const unsigned TF = GSR_FLAG_COMPOSITE_TILES;
        rc = gsr_forward(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color, radii, ws, nbytes, cap, TF, st);
        if (rc == GPSGS_OK) rc = gsr_backward(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, radii, dgp, g3, g2, gc, go, gs, gr, ws, nbytes, cap, TF, st);
        if (rc != GPSGS_OK) { printf("tile-family baseline rc=%d\n", rc); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(img.data(), color, img.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgc.data(), gc, 12 * P, hipMemcpyDeviceToHost));
        GsrViewExt ext = {};
        ext.row_range = drange;
        ext.order_hint = h.max_tile_count;
        rc = gsr_forward_ex(Pcap, W, H, wm, wc, wo, wsc, wr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color2, radii2, ws2, nb2, cap, TF, st, nullptr, 0u, &ext);
        if (rc != GPSGS_OK) { printf("gsr_forward_ex rc=%d\n", rc); return 1; }
        GsrHeader h2;
        if (gsr_read_header(ws2, &h2, st) != GPSGS_OK) return 1;
        if (h2.overflow || h2.num_rendered != h.num_rendered || h2.num_points != (uint32_t)P || h2.row_overflow) {
            printf("row-range header: overflow=%u R=%llu points=%u\n", h2.overflow, (unsigned long long)h2.num_rendered, h2.num_points); return 1;
        }
        std::vector<float> img2(3 * W * H);
        CK(hipMemcpy(img2.data(), color2, img2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < img.size(); k++) if (img2[k] != img[k]) { printf("row-range image differs at %zu\n", k); return 1; }
        float *w3, *w2, *wgc, *wgo, *wgs, *wgr;
        CK(hipMalloc(&w3, 36 * P)); CK(hipMalloc(&w2, 36 * P)); CK(hipMalloc(&wgc, 36 * P)); CK(hipMalloc(&wgo, 12 * P)); CK(hipMalloc(&wgs, 36 * P)); CK(hipMalloc(&wgr, 48 * P));
        for (int pass = 0; pass < 2; pass++) {  // 0: all gradients, 1: without the colour gradient
            rc = gsr_backward_ex(Pcap, W, H, wm, wc, wo, wsc, wr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, radii2, dgp, w3, w2, wgc, wgo, wgs, wgr, ws2, nb2, cap,
                                 TF | (pass ? GSR_FLAG_NO_COLOR_GRAD : 0u), st, &ext);
            if (rc != GPSGS_OK) { printf("gsr_backward_ex rc=%d\n", rc); return 1; }
            CK(hipStreamSynchronize(st));
            std::vector<float> a(3 * P), b(3 * P), col(3 * P);
            CK(hipMemcpy(a.data(), g3, 12 * P, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), w3 + 3 * P, 12 * P, hipMemcpyDeviceToHost));   // rows [P, 2P) of the batch-wide gradient array
            CK(hipMemcpy(col.data(), wgc + 3 * P, 12 * P, hipMemcpyDeviceToHost));
            for (int k = 0; k < 3 * P; k++) {
                if (a[k] != b[k]) { printf("row-range dL_dmeans3D differs at %d (pass %d)\n", k, pass); return 1; }
                if (pass == 0 ? col[k] != hgc[k] : col[k] != 0.f) { printf("row-range dL_dcolors wrong at %d (pass %d)\n", k, pass); return 1; }
            }
        }
        // GSR_FLAG_WAVE_PRIORITY is a scheduling hint: forward and backward with it must reproduce every bit
        {
            const unsigned PF = TF | GSR_FLAG_WAVE_PRIORITY;
            rc = gsr_forward_ex(Pcap, W, H, wm, wc, wo, wsc, wr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color2, radii2, ws2, nb2, cap, PF, st, nullptr, 0u, &ext);
            if (rc == GPSGS_OK)
                rc = gsr_backward_ex(Pcap, W, H, wm, wc, wo, wsc, wr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, radii2, dgp, w3, w2, wgc, wgo, wgs, wgr, ws2, nb2, cap, PF, st, &ext);
            if (rc != GPSGS_OK) { printf("wave-priority pass rc=%d\n", rc); return 1; }
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(img2.data(), color2, img2.size() * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < img.size(); k++) if (img2[k] != img[k]) { printf("wave-priority image differs at %zu\n", k); return 1; }
            std::vector<float> a(3 * P), b(3 * P), col(3 * P);
            CK(hipMemcpy(a.data(), g3, 12 * P, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), w3 + 3 * P, 12 * P, hipMemcpyDeviceToHost));
            CK(hipMemcpy(col.data(), wgc + 3 * P, 12 * P, hipMemcpyDeviceToHost));
            for (int k = 0; k < 3 * P; k++)
                if (a[k] != b[k] || col[k] != hgc[k]) { printf("wave-priority gradients differ at %d\n", k); return 1; }
        }
        // ABI version 4: DIRECT bin lists (GsrViewExt.bin_capacity): the workspace is sized with gsr_workspace_bytes_ex, forward and backward carry the
        // same capacity; image, header totals and gradients must equal the scanned-list baseline bit for bit.  A capacity the longest list does not
        // fit must come back as an overflow whose header says so (max_tile_count > capacity), and nothing is rendered.
        {
            GsrViewExt dx = {};
            dx.bin_capacity = 512;
            if (!gsr_direct_lists_ok(W, H, 512) || gsr_direct_lists_ok(W, H, 100) || gsr_direct_lists_ok(W, H, 2048)) { printf("gsr_direct_lists_ok\n"); return 1; }
            const size_t nb3 = gsr_workspace_bytes_ex(P, W, H, cap, 512, 0);
            if (nb3 == 0 || gsr_workspace_bytes_ex(P, W, H, cap, 0, 0) != nbytes) { printf("gsr_workspace_bytes_ex\n"); return 1; }
            void *ws3; CK(hipMalloc(&ws3, nb3));
            rc = gsr_forward_ex(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color2, radii2, ws3, nb3, cap, TF, st, nullptr, 0u, &dx);
            if (rc == GPSGS_OK) rc = gsr_backward_ex(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, radii2, dgp, w3, w2, wgc, wgo, wgs, wgr, ws3, nb3, cap, TF, st, &dx);
            if (rc != GPSGS_OK) { printf("direct lists rc=%d\n", rc); return 1; }
            GsrHeader h3;
            if (gsr_read_header(ws3, &h3, st) != GPSGS_OK) return 1;
            if (h3.overflow || h3.num_rendered != h.num_rendered || h3.max_tile_count != h.max_tile_count || h3.num_busy_wgs != h.num_busy_wgs || h3.num_slots != h.num_slots) {
                printf("direct-list header differs: overflow=%u R=%llu longest=%u busy=%u slots=%u\n", h3.overflow, (unsigned long long)h3.num_rendered, h3.max_tile_count,
                       h3.num_busy_wgs, h3.num_slots);
                return 1;
            }
            CK(hipMemcpy(img2.data(), color2, img2.size() * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < img.size(); k++) if (img2[k] != img[k]) { printf("direct-list image differs at %zu\n", k); return 1; }
            std::vector<float> a(3 * P), b(3 * P), col(3 * P);
            CK(hipMemcpy(a.data(), g3, 12 * P, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), w3, 12 * P, hipMemcpyDeviceToHost));
            CK(hipMemcpy(col.data(), wgc, 12 * P, hipMemcpyDeviceToHost));
            for (int k = 0; k < 3 * P; k++)
                if (a[k] != b[k] || col[k] != hgc[k]) { printf("direct-list gradients differ at %d\n", k); return 1; }
            if (h.max_tile_count > 64) {  // a capacity of 64 entries per bin cannot hold this view's longest list
                dx.bin_capacity = 64;
                const size_t nb4 = gsr_workspace_bytes_ex(P, W, H, cap, 64, 0);
                if (nb4 == 0 || nb4 > nb3) { printf("workspace for 64-entry bins: %zu\n", nb4); return 1; }
                rc = gsr_forward_ex(P, W, H, dm, dc, dop, ds, dr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color2, radii2, ws3, nb4, cap, TF, st, nullptr, 0u, &dx);
                if (rc != GPSGS_OK || gsr_read_header(ws3, &h3, st) != GPSGS_OK) return 1;
                if (!h3.overflow || h3.max_tile_count != h.max_tile_count || h3.num_rendered != h.num_rendered) { printf("too long a list for direct bins was not reported\n"); return 1; }
            }
            (void)hipFree(ws3);
        }
        // a range longer than the capacity is reported like an overflow
        std::vector<uint32_t> big = {0u, (uint32_t)(3 * P)};
        uint32_t *dbig = dev(big);
        ext.row_range = dbig;
        rc = gsr_forward_ex(Pcap, W, H, wm, wc, wo, wsc, wr, 1.0f, W / (2 * fx), H / (2 * fx), dv, dp, dbg, color2, radii2, ws2, nb2, cap, TF, st, nullptr, 0u, &ext);
        if (rc != GPSGS_OK || gsr_read_header(ws2, &h2, st) != GPSGS_OK) return 1;
        if (!h2.overflow || !h2.row_overflow || h2.num_points != (uint32_t)(3 * P)) { printf("row overflow not reported\n"); return 1; }
    }
    // argument checking happens before any launch
    if (gsr_forward(P, 0, H, dm, dc, dop, ds, dr, 1.f, 1.f, 1.f, dv, dp, dbg, color, radii, ws, nbytes, cap, 0, st) != GPSGS_E_INVALID) return 1;
    if (gsr_forward(P, W, H, dm, dc, dop, ds, dr, 1.f, 1.f, 1.f, dv, dp, dbg, color, radii, ws, 64, cap, 0, st) != GPSGS_E_WORKSPACE) return 1;
    printf("CAPI_HOST_OK R=%llu longest_bin=%u image_sum=%.3f grad_sum=%.3f\n", (unsigned long long)h.num_rendered, h.max_tile_count, sum, gsum);
    return 0;
}
