// host_shim.cpp -- the per-call HOST work of the drop-in rasteriser module, compiled (lib/_gpsgs_fast.so, a CPython extension).
//
// What it replaces: rasterizer.py's _RasterizeGaussians (a Python torch.autograd.Function) + _forward_impl / _backward_impl for the call shape the
// reference uses -- GaussianRasterizer(raster_settings)(means3D=..., means2D=..., colors_precomp=..., opacities=..., scales=..., rotations=...) and
// image.backward() (/root/reference/gaussian_renderer/__init__.py:36-62, train_stage2.py:83) -- in GPSGS_CHECK=sync mode.  Round 5 measured ~240 us
// of Python / PyTorch host work per ~250 us GPU step on that path (profiles/r05_host_timeline.md): the contract number depended on pinning the
// process to one L3 domain.  Here the same steps -- argument checks, three allocations, gsr_forward_ex with the early capacity notification, the
// spin on the pinned header word, the autograd node, six gradient tensors carved out of one buffer, gsr_backward_ex -- are a C++
// torch::autograd::Function that calls the SAME C-ABI (include/gpsgs.h).  No kernel lives here; nothing is computed here.
//
// What stays in Python (rasterizer.py): the capacity policy state and every slow path -- an overflow (the call is simply repeated through the Python
// path, which repairs it), deferred / unchecked modes, graph capture, row-range batches, SH colours, precomputed covariances, debug / timing flags,
// tensors that need conversion.  rasterize() returns None for anything it does not take, and the caller falls through.
#include <torch/extension.h>

#include <ATen/hip/HIPContext.h>
#include <torch/csrc/autograd/custom_function.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "../../include/gpsgs.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---- pinned notification slots (gsr_forward_ex's host_header_out): 64 per device, handed out with in-flight accounting -------------------------------
struct Ring {
    Tensor pinned;  // [64, 8] int32, pinned host memory
    std::vector<int> free_slots;
    uint32_t seq = 0;
};
std::mutex g_mu;
Ring g_ring[64];

bool acquire_slot(int dev, int &slot, volatile uint32_t *&w32, uint32_t &seq) {
    std::lock_guard<std::mutex> g(g_mu);
    Ring &r = g_ring[dev];
    if (!r.pinned.defined()) {
        r.pinned = at::zeros({64, 8}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
        for (int k = 63; k >= 0; k--) r.free_slots.push_back(k);
    }
    if (r.free_slots.empty()) return false;
    slot = r.free_slots.back();
    r.free_slots.pop_back();
    seq = r.seq = r.seq % 0x7fffffffu + 1u;  // never 0, never what the slot holds from its previous use
    w32 = reinterpret_cast<volatile uint32_t *>(r.pinned.data_ptr<int>()) + 8 * slot;
    return true;
}
void release_slot(int dev, int slot) {
    std::lock_guard<std::mutex> g(g_mu);
    g_ring[dev].free_slots.push_back(slot);
}

struct Header {
    long long R = 0, nbytes = 0;
    unsigned overflow = 0, longest = 0, slots = 0, npts = 0;
    double wait_us = 0.0;  // time spent spinning on the notification word (the GPU's share of the forward call: tools/host_time.py subtracts it)
};
thread_local Header t_hdr;

inline bool f32_cuda_contig(const Tensor &t, const c10::Device &dev) {
    return t.defined() && t.is_cuda() && t.device() == dev && t.scalar_type() == at::kFloat && t.is_contiguous();
}

struct Rasterize : public torch::autograd::Function<Rasterize> {
    static variable_list forward(AutogradContext *ctx, Tensor means3D, Tensor means2D, Tensor colors, Tensor opacities, Tensor scales, Tensor rotations,
                                 Tensor bg, Tensor view, Tensor proj, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                                 int64_t cap, int64_t bin_cap, int64_t order_hint, int64_t flags, bool needs_grad) {
        (void)means2D;  // takes part only as the holder of dL/dmean2D (the reference retains its gradient: gaussian_renderer/__init__.py:24-29)
        const int P = (int)means3D.size(0);
        const c10::Device dev = means3D.device();
        const int di = dev.index();
        const at::DeviceGuard guard(dev);
        auto stream = at::hip::getCurrentHIPStream(di);
        const auto f32 = means3D.options();
        Tensor color = at::empty({3, H, W}, f32);
        Tensor radii = at::empty({P}, f32.dtype(at::kInt));
        const size_t nbytes = gsr_workspace_bytes_ex(P, (int)W, (int)H, cap, (uint32_t)bin_cap, needs_grad ? 0 : 1);
        TORCH_CHECK(nbytes > 0, "gps_gaussian_amd: invalid workspace dimensions");
        Tensor ws = at::empty({(int64_t)nbytes}, f32.dtype(at::kByte));
        int slot = -1;
        volatile uint32_t *w32 = nullptr;
        uint32_t seq = 0;
        TORCH_CHECK(acquire_slot(di, slot, w32, seq), "gps_gaussian_amd: no free notification slot");
        GsrViewExt ext = {};
        ext.order_hint = (uint32_t)order_hint;
        ext.bin_capacity = (uint32_t)bin_cap;
        const int rc = gsr_forward_ex(P, (int)W, (int)H, means3D.data_ptr<float>(), colors.data_ptr<float>(), opacities.data_ptr<float>(), scales.data_ptr<float>(),
                                      rotations.data_ptr<float>(), (float)scale_modifier, (float)tanfovx, (float)tanfovy, view.data_ptr<float>(), proj.data_ptr<float>(),
                                      bg.data_ptr<float>(), color.data_ptr<float>(), radii.data_ptr<int>(), ws.data_ptr(), nbytes, cap, (unsigned)flags,
                                      (void *)stream.stream(), (void *)w32, seq, &ext);
        if (rc != GPSGS_OK) {
            release_slot(di, slot);
            TORCH_CHECK(false, "gps_gaussian_amd: gsr_forward_ex failed (", rc, ")");
        }
        // the device publishes the header to the pinned slot right after the binning: the host checks capacity while sort / compositing still run
        unsigned long n = 0;
        bool lost = false;
        const auto t_wait0 = std::chrono::steady_clock::now();
        while (w32[7] != seq) {
            if ((++n & 0x3fffu) == 0 && stream.query()) {  // the stream drained without the store: surface it instead of spinning for ever
                if (w32[7] == seq) break;
                lost = true;
                break;
            }
        }
        Header h;
        h.wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_wait0).count();
        h.R = (long long)((unsigned long long)w32[0] | ((unsigned long long)w32[1] << 32));
        h.overflow = w32[2]; h.longest = w32[3]; h.slots = w32[5]; h.npts = w32[6];
        h.nbytes = (long long)nbytes;
        release_slot(di, slot);
        TORCH_CHECK(!lost, "gps_gaussian_amd: the rasteriser forward finished without publishing its header");
        t_hdr = h;
        ctx->save_for_backward({means3D, colors, opacities, scales, rotations, view, proj, bg, radii, ws});
        ctx->saved_data["H"] = H; ctx->saved_data["W"] = W;
        ctx->saved_data["tanfovx"] = tanfovx; ctx->saved_data["tanfovy"] = tanfovy; ctx->saved_data["scale_modifier"] = scale_modifier;
        ctx->saved_data["cap"] = cap; ctx->saved_data["bin_cap"] = bin_cap; ctx->saved_data["flags"] = flags;
        ctx->mark_non_differentiable({radii, ws});
        ctx->set_materialize_grads(false);
        return {color, radii, ws};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_outputs) {
        variable_list out(19);  // one entry per forward argument; undefined = None
        Tensor g = grad_outputs[0];
        if (!g.defined()) return out;  // the image did not take part in the loss
        const auto saved = ctx->get_saved_variables();
        const Tensor &m3 = saved[0], &col = saved[1], &opa = saved[2], &sca = saved[3], &rot = saved[4], &view = saved[5], &proj = saved[6], &bg = saved[7],
                     &radii = saved[8], &ws = saved[9];
        const int64_t H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        const int64_t cap = ctx->saved_data["cap"].toInt(), bin_cap = ctx->saved_data["bin_cap"].toInt(), flags = ctx->saved_data["flags"].toInt();
        const double tanfovx = ctx->saved_data["tanfovx"].toDouble(), tanfovy = ctx->saved_data["tanfovy"].toDouble(), smod = ctx->saved_data["scale_modifier"].toDouble();
        const int P = (int)m3.size(0);
        const c10::Device dev = m3.device();
        const at::DeviceGuard guard(dev);
        if (g.scalar_type() != at::kFloat || !g.is_contiguous()) g = g.to(at::kFloat).contiguous();  // H3: may arrive non-contiguous / in another dtype
        // one allocation, six contiguous gradient arrays carved out of it (quaternion gradient first: it is stored as float4)
        Tensor buf = at::empty({(int64_t)P * 17}, m3.options());
        int64_t o = 0;
        auto carve = [&](int64_t c) { Tensor t = buf.narrow(0, o, (int64_t)P * c).view({P, c}); o += (int64_t)P * c; return t; };
        Tensor d_rot = carve(4), d_m3 = carve(3), d_m2 = carve(3), d_col = carve(3), d_sc = carve(3), d_op = carve(1);
        const bool color_grad = ctx->needs_input_grad(2);  // stage 2 never differentiates the colours (input pixels): the backward then skips their sums
        GsrViewExt ext = {};
        ext.bin_capacity = (uint32_t)bin_cap;
        auto stream = at::hip::getCurrentHIPStream(dev.index());
        const int rc = gsr_backward_ex(P, (int)W, (int)H, m3.data_ptr<float>(), col.data_ptr<float>(), opa.data_ptr<float>(), sca.data_ptr<float>(), rot.data_ptr<float>(),
                                       (float)smod, (float)tanfovx, (float)tanfovy, view.data_ptr<float>(), proj.data_ptr<float>(), bg.data_ptr<float>(), radii.data_ptr<int>(),
                                       g.data_ptr<float>(), d_m3.data_ptr<float>(), d_m2.data_ptr<float>(), d_col.data_ptr<float>(), d_op.data_ptr<float>(),
                                       d_sc.data_ptr<float>(), d_rot.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), cap,
                                       (unsigned)flags | (color_grad ? 0u : GSR_FLAG_NO_COLOR_GRAD), (void *)stream.stream(), &ext);
        TORCH_CHECK(rc == GPSGS_OK, "gps_gaussian_amd: gsr_backward_ex failed (", rc, ")");
        out[0] = d_m3; out[1] = d_m2;
        if (color_grad) out[2] = d_col;
        out[3] = opa.dim() == 2 ? d_op : d_op.view({P});
        out[4] = d_sc; out[5] = d_rot;
        return out;
    }
};

// -> (color, radii, workspace) or None when the call is not the fast path's (the caller then takes the Python path).  The header of the forward --
// what the capacity policy learns from -- is read with last_header().
py::object rasterize(const Tensor &means3D, const Tensor &means2D, const Tensor &colors, const Tensor &opacities, const Tensor &scales, const Tensor &rotations,
                     const Tensor &bg, const Tensor &view, const Tensor &proj, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t cap,
                     int64_t bin_cap, int64_t order_hint, int64_t flags) {
    if (!means3D.defined() || !means3D.is_cuda() || means3D.dim() != 2 || means3D.size(1) != 3) return py::none();
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    if (P <= 0 || P > 0x7fffffffLL || dev.index() < 0 || dev.index() >= 64 || H <= 0 || W <= 0) return py::none();
    if (!f32_cuda_contig(means3D, dev) || !f32_cuda_contig(colors, dev) || !f32_cuda_contig(opacities, dev) || !f32_cuda_contig(scales, dev) || !f32_cuda_contig(rotations, dev) ||
        !f32_cuda_contig(bg, dev) || !f32_cuda_contig(view, dev) || !f32_cuda_contig(proj, dev))
        return py::none();
    if (colors.dim() != 2 || colors.size(0) != P || colors.size(1) != 3 || scales.dim() != 2 || scales.size(0) != P || scales.size(1) != 3 || rotations.dim() != 2 ||
        rotations.size(0) != P || rotations.size(1) != 4 || opacities.numel() != P || bg.numel() != 3 || view.numel() != 16 || proj.numel() != 16)
        return py::none();
    if (reinterpret_cast<uintptr_t>(rotations.data_ptr()) % 16) return py::none();  // the quaternions are read as float4
    if (bg.requires_grad() || view.requires_grad() || proj.requires_grad()) return py::none();
    if (!means2D.defined() || means2D.dim() != 2 || means2D.size(0) != P || means2D.size(1) != 3 || !means2D.is_cuda()) return py::none();
    const bool needs_grad = at::GradMode::is_enabled() && (means3D.requires_grad() || means2D.requires_grad() || colors.requires_grad() || opacities.requires_grad() ||
                                                           scales.requires_grad() || rotations.requires_grad());
    variable_list out;
    {
        py::gil_scoped_release nogil;  // the spin on the notification word must not hold the interpreter (the autograd thread needs it)
        out = Rasterize::apply(means3D, means2D, colors, opacities, scales, rotations, bg, view, proj, H, W, tanfovx, tanfovy, scale_modifier, cap, bin_cap, order_hint, flags,
                               needs_grad);
    }
    return py::make_tuple(out[0], out[1], out[2]);
}

py::tuple last_header() {
    const Header &h = t_hdr;
    return py::make_tuple(h.R, h.overflow, h.longest, h.slots, h.npts, h.nbytes, h.wait_us);
}

int slots_in_flight(int dev) {
    std::lock_guard<std::mutex> g(g_mu);
    if (dev < 0 || dev >= 64 || !g_ring[dev].pinned.defined()) return 0;
    return 64 - (int)g_ring[dev].free_slots.size();
}

}  // namespace

PYBIND11_MODULE(_gpsgs_fast, m) {
    m.doc() = "compiled host path of gps_gaussian_amd.rasterizer (sync mode, the reference's call shape); calls libgpsgs_hip.so's C-ABI";
    m.def("rasterize", &rasterize);
    m.def("last_header", &last_header);
    m.def("slots_in_flight", &slots_in_flight);
    m.def("abi_version", []() { return gpsgs_abi_version(); });
}
