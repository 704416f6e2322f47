"""bench.py -- the hot-path benchmark the driver runs.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric (BASELINE.json): novel views/s at 1024x1024 with ~600k pixel-Gaussians.  One "step" = one pass of the render
hot path over one synthetic view of BASELINE config 2: HIP rasteriser forward + backward (1024x1024, P = 600,000
Gaussians resident in HBM before the timed region), called the way the reference calls it -- GaussianRasterizer(raster_settings)(means3D=..., ...)
+ image.backward(), /root/reference/gaussian_renderer/__init__.py:51-62 -- one view at a time with the exact capacity check: `value`.
What a caller that owns its buffers gets from the C-ABI sessions (one view at a time, and six views in flight on six streams) is reported
next to it (`session`), as are forward-only throughput, the stage-2 gradient set and the other BASELINE configurations.
N > 1: every rank renders its own views (independent units, no data-path collective) -> weak scaling; the only
collectives are the timing barrier and the MAX-over-ranks reduction of the elapsed time.

The printed JSON line also carries
  roofline     : the dominant kernel (backward compositing) priced against the HBM roofline with ALGORITHMIC bytes
                 (DESIGN.md "Kernels and rooflines"); its duration is measured live with hipEvents on the launch stream;
  cpu_baseline : the fp32 CPU oracle ("port": the reference's rasteriser is an external CUDA extension that cannot be
                 built here) on the host cores, rank 0 at N=1 only, on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_PEAK_TLANEOPS = 78.6   # 157.3 TFLOP/s fp32 vector = 78.6 T FMA lane-ops/s


def stage_bytes(P, R, NB, npix, n_slots=None, n_rec=None, direct=False, color_grad=True):
    """ALGORITHMIC bytes per launch of every stage (SURVEY.md section 8d convention: each input read once, each intermediate
    written once and read once, each output written once).  R = measured (Gaussian, 8x8-bin) instances, NB = bins, n_slots = gradient-record slots
    (bin-rect cells of all Gaussians), n_rec = slots the compositing backward wrote a record into (<= R).  Unknown counts fall back to R.
    direct: direct bin lists (round 6) -- no scan launch: the scatter needs no offsets, workgroups of its launch do what is left of the scan.
      preprocess      44 P in + 48 P splat record + 16 P bin record + 4 P radii + 8 P slot prefix          = 120 P
      scan            8 NB counters in (two arrays) + 8 NB offsets / cursors + 4 NB work order            = 20 NB        (scanned lists only)
      scatter         16 P bin records + 8 R keys (+ direct: 8 NB counters in + 4 NB work order)           = 16 P + 8 R (+ 12 NB)
      sort            8 R keys in + 4 R ids out + 8 NB ranges / counts + work order                        = 12 R + 8 NB  (direct lists + tile family: no such launch --
                      the forward compositing waves sort their own lists, and these bytes are the forward's)
      composite_fwd   4 R ids + 36 R splat record fields + 8 NB + 12 Npix image + 8 Npix state + n_slots flag bytes cleared
      composite_bwd   as the forward's reads + 12 Npix dL/dpix + 8 Npix state, + 37 B per record written (33 without colour sums: one sector + flag)
      preprocess_bwd  n_slots flags + 36 n_rec records (32 without colour sums) + 40 P inputs + 8 P slot prefix + 68 P gradients"""
    n_slots = R if n_slots is None else n_slots
    n_rec = R if n_rec is None else n_rec
    rec_w, rec_r = (37, 36) if color_grad else (33, 32)
    fused_sort = direct and os.environ.get("GPSGS_COMPOSITE", "tiles") == "tiles" and os.environ.get("GPSGS_FUSED_SORT", "1") != "0"
    out = {"preprocess": 120 * P, "scatter": 16 * P + 8 * R + (12 * NB if direct else 0),
           "composite_fwd": 40 * R + 8 * NB + 20 * npix + n_slots + (12 * R if fused_sort else 0), "composite_bwd": 40 * R + 8 * NB + 20 * npix + rec_w * n_rec,
           "preprocess_bwd": n_slots + rec_r * n_rec + 116 * P}
    if not fused_sort:
        out.update(sort=12 * R + 8 * NB)
    if not direct:
        out.update(scan=20 * NB)
    return out


def stage_bound(us, alg_bytes, counter_bytes, valu_instr, sclk_mhz):
    """What bounds a kernel, from its duration and the separate counter passes: `valu_issue_frac` = wave64 VALU instructions x 4
    cycles / (1,024 SIMDs x shader clock x t) -- an upper bound of the issue-port use --, `hbm_frac` from the COUNTER bytes when they are known (the
    algorithmic bytes of a kernel whose re-reads are served by the L2 / Infinity Cache overstate what the HBM sees: the regime's forward compositing
    read 0.60 by formula, 0.11 by counters), and the name of the larger of the two when it is at least a half -- else "latency" (dependent round
    trips / occupancy: neither pipe is busy)."""
    t = us * 1e-6
    out = {"hbm_frac_algorithmic": round(alg_bytes / t / 1e9 / HBM_PEAK_GBS, 5)}
    hb = out["hbm_frac_algorithmic"]
    if counter_bytes:
        out["hbm_frac_counters"] = hb = round(counter_bytes / t / 1e9 / HBM_PEAK_GBS, 5)
    out["hbm_frac"] = hb
    vf = None
    if valu_instr:
        vf = out["valu_issue_frac"] = round(valu_instr * 4.0 / (1024 * (sclk_mhz or 2400.0) * 1e6 * t), 4)
    out["bound"] = "valu issue" if (vf is not None and vf >= 0.5 and vf >= hb) else "hbm" if hb >= 0.5 else "latency (dependent round trips / occupancy)"
    return out


def neighbours_leg(dev, B=4, res=1024, render_res=2048):
    """The kernels either side of the rasteriser at BASELINE config 4's sizes, each with its ALGORITHMIC bytes, its GPU
    time and the HBM fraction that gives (until round 5 only speed-ups over eager PyTorch were reported).  Timing: the calls are
    queued behind a ~2 ms spin kernel so that the GPU runs them back to back, bracketed by two events (no host gaps inside the bracket); a call is
    one to three launches (the kernels are named).  Backward = (forward + backward) - forward."""
    import torch
    from gps_gaussian_amd import corr as K, loss as L, pack as PK, unproject as U
    out = {}

    def gpu_us(fn, n=10):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(5_000_000)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n * 1e3

    def row(name, kernels, fwd_us, fb_us, fwd_bytes, bwd_bytes, note=""):
        bwd_us = max(fb_us - fwd_us, 1e-3)
        out[name] = {"kernels": kernels, "forward": {"us": round(fwd_us, 1), "algorithmic_bytes": int(fwd_bytes), "hbm_frac": round(fwd_bytes / (fwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                     "backward": {"us": round(bwd_us, 1), "algorithmic_bytes": int(bwd_bytes), "hbm_frac": round(bwd_bytes / (bwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
        if note:
            out[name]["note"] = note

    torch.manual_seed(0)
    # f2: fused L1 + SSIM on the rendered batch [B, 3, 2048, 2048]
    pred = torch.rand(B, 3, render_res, render_res, device=dev, requires_grad=True)
    gt = torch.rand(B, 3, render_res, render_res, device=dev)
    npl = pred.numel()

    def loss_f():
        with torch.no_grad():
            L.l1_and_ssim(pred, gt)

    def loss_fb():
        pred.grad = None
        l1, ss = L.l1_and_ssim(pred, gt)
        (0.8 * l1 + 0.2 * (1 - ss)).backward()
    row("f2_l1_ssim_loss", "k_loss_fwd / k_loss_bwd (csrc/fused_loss.hip)", gpu_us(loss_f), gpu_us(loss_fb), 20 * npl, 24 * npl,
        "forward: pred + gt in, three derivative maps out; backward: those five in, d_pred out (+ two scalar torch kernels for the weighted sum)")
    del pred, gt
    # f3: disparity -> inverse depth -> world points, 2 B views of 1024^2
    V, S = 2 * B, res
    flow = torch.rand(V, 1, S, S, device=dev, requires_grad=True)
    mask = (torch.rand(V, 1, S, S, device=dev) > 0.3).float()
    intr = torch.tensor([[800.0, 0, S / 2], [0, 800.0, S / 2], [0, 0, 1]], device=dev)[None].repeat(V, 1, 1)
    extr = torch.eye(4, device=dev)[:3][None].repeat(V, 1, 1).contiguous()
    tf = torch.full((V,), 120.0, device=dev)
    gx = torch.randn(V, S * S, 3, device=dev)

    def unp_f():
        with torch.no_grad():
            U.unproject(flow, mask, intr, intr, extr, tf)

    def unp_fb():
        flow.grad = None
        d, xyz, _ = U.unproject(flow, mask, intr, intr, extr, tf)
        xyz.backward(gx)
    npx = V * S * S
    row("f3_unproject", "k_unproject_fwd / k_unproject_bwd (csrc/unproject.hip)", gpu_us(unp_f), gpu_us(unp_fb), 25 * npx, 24 * npx)
    # f1: mask compaction + pack of B stereo pairs
    data = {}
    for v in ("lmain", "rmain"):
        data[v] = dict(xyz=torch.randn(B, S * S, 3, device=dev, requires_grad=True), img=torch.rand(B, 3, S, S, device=dev),
                       rot_maps=torch.randn(B, 4, S, S, device=dev, requires_grad=True), scale_maps=torch.rand(B, 3, S, S, device=dev, requires_grad=True),
                       opacity_maps=torch.rand(B, 1, S, S, device=dev, requires_grad=True), pts_valid=(torch.rand(B, S * S, device=dev) < 0.286))
    rows_n = int(sum(int(data[v]["pts_valid"].sum()) for v in data))

    def pack_f():
        with torch.no_grad():
            PK.pack_views(data)

    def pack_fb():
        for v in data:
            for k in ("xyz", "rot_maps", "scale_maps", "opacity_maps"):
                data[v][k].grad = None
        o = PK.pack_views(data)
        (o[0].sum() + o[2].sum() + o[3].sum() + o[4].sum()).backward()
    row("f1_pack_views", "k_pack_count / k_pack_scan / k_pack_write, k_pack_bwd (csrc/pack_views.hip)", gpu_us(pack_f), gpu_us(pack_fb),
        61 * npx + 56 * rows_n, 48 * npx + 56 * rows_n, "%d of %d pixels valid; the backward leg includes four torch sum-backward fills" % (rows_n, npx))
    del data
    # f4: correlation volume + pyramid, lookup of all levels, convex upsampling (RAFT-Stereo shapes for B pairs at 1024^2)
    N, D, H, W, f = 2 * B, 192, res // 8, res // 8, 8
    f1 = torch.randn(N, D, H, W, device=dev, requires_grad=True)
    f2 = torch.randn(N, D, H, W, device=dev, requires_grad=True)
    coords = torch.rand(N, 2, H, W, device=dev) * W
    pyr_b = 4 * N * H * W * (W + W // 2 + W // 4 + W // 8)
    blk = [None]

    def cv_f():
        with torch.no_grad():
            blk[0] = K.CorrBlockFast1D(f1, f2, num_levels=4, radius=4)

    def cv_fb():
        f1.grad = f2.grad = None
        b_ = K.CorrBlockFast1D(f1, f2, num_levels=4, radius=4)
        sum(v.sum() for v in b_.volumes).backward()
    row("f4_corr_volume_pyramid", "k_cv_build / k_cv_build_bwd (csrc/corr_pyramid.hip, fp32 MFMA)", gpu_us(cv_f), gpu_us(cv_fb), 2 * 4 * N * D * H * W + pyr_b,
        pyr_b + 4 * 4 * N * D * H * W, "2 N H W1 W2 D = %.1f GFLOP per direction on the fp32 matrix path as well (157 TFLOP/s peak)" % (2e-9 * N * H * W * W * D))
    cv_f()

    def lk_f():
        with torch.no_grad():
            blk[0](coords)
    us_lk = gpu_us(lk_f)
    out["f4_lookup_all_levels"] = {"kernels": "k_cs_lookup (one launch for four levels)", "forward": {"us": round(us_lk, 1), "algorithmic_bytes": 4 * N * H * W * (1 + 36 + 4 * 2 * 9),
                                                                                                     "hbm_frac": round(4 * N * H * W * (1 + 36 + 72) / (us_lk * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                                   "note": "gathers 2 x 9 taps per level from a pyramid that sits in the Infinity Cache: latency-bound, listed for completeness"}
    mask_u = torch.randn(N, 9 * f * f, H, W, device=dev, requires_grad=True)
    flw = torch.randn(N, 2, H, W, device=dev, requires_grad=True)

    def up_f():
        with torch.no_grad():
            K.upsample_flow(flw, mask_u, f)

    def up_fb():
        mask_u.grad = flw.grad = None
        K.upsample_flow(flw, mask_u, f).sum().backward()
    nm, no = mask_u.numel(), N * 2 * f * H * f * W
    row("f4_convex_upsample", "k_up_fwd / k_up_bwd (csrc/corr_pyramid.hip)", gpu_us(up_f), gpu_us(up_fb), 4 * (nm + flw.numel() + no), 4 * (2 * nm + no + 2 * flw.numel()))
    torch.cuda.empty_cache()
    return out


def counter_traffic(fname, P, W, H, R):
    """Per-stage HBM traffic from the separate rocprofv3 --pmc passes (tools/prof_r06.sh -> profiles/<fname>, which records the workload it was
    measured on); {} unless THIS run is that workload (same P, size, R within 1 %)."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", fname)))
        wl = pj.get("workload", {})
        if wl.get("P") == P and wl.get("W") == W and wl.get("H") == H and abs(wl.get("R", -1) - R) <= 0.01 * R:
            return pj
    except Exception:  # noqa: BLE001
        pass
    return {}


def stage2_leg(args, sample, dev, rank, local_rank, world, D, timed, batch=4, steps=10):
    """Secondary measurement (never part of `value`): the non-network part of one stage-2 training iteration
    (/root/reference/train_stage2.py:65-72,83) with the reference's batch of 4 stereo pairs per GPU: pts2render (fused
    mask-compaction + pack, then one raster forward per pair) -> 0.8 L1 + 0.2 (1 - SSIM) -> backward down to the per-pixel
    Gaussian maps, followed by the ONE exchange step of data-parallel stage-2 training: a mean all-reduce of the network's
    5,144,408 fp32 gradients (20.6 MB, one RCCL bucket; no-op on one GPU).  The RAFT-Stereo / regressor networks themselves
    are out of scope and not executed.  Every rank always reaches every collective: a local failure is
    agreed on through a MIN all-reduce before any timed collective is issued."""
    import torch
    import torch.distributed as dist
    from gps_gaussian_amd import loss as L
    from gps_gaussian_amd import render_api
    ok, err, step = 1, None, None
    try:
        B, res = batch, args.res
        rres = args.render_res or args.res
        cam = sample["novel_view"]
        data = {}
        for v in ("lmain", "rmain"):
            d = sample[v]
            rep = lambda a, *r: torch.from_numpy(a).to(dev)[None].repeat(B, *r)
            data[v] = dict(img=rep(d["img"], 1, 1, 1), xyz=rep(d["xyz"], 1, 1).requires_grad_(True), pts_valid=rep(d["pts_valid"], 1),
                           rot_maps=rep(d["rot_maps"], 1, 1, 1).requires_grad_(True), scale_maps=rep(d["scale_maps"], 1, 1, 1).requires_grad_(True),
                           opacity_maps=rep(d["opacity_maps"], 1, 1, 1).requires_grad_(True))
        data["novel_view"] = dict(FovX=torch.tensor([float(cam["FovX"])] * B), FovY=torch.tensor([float(cam["FovY"])] * B),
                                  width=torch.tensor([rres] * B), height=torch.tensor([rres] * B),
                                  world_view_transform=torch.from_numpy(cam["world_view_transform"])[None].repeat(B, 1, 1).pin_memory(),
                                  full_proj_transform=torch.from_numpy(cam["full_proj_transform"])[None].repeat(B, 1, 1).pin_memory(),
                                  camera_center=torch.from_numpy(cam["camera_center"])[None].repeat(B, 1))
        gt = torch.rand(B, 3, rres, rres, device=dev)
        net = [torch.nn.Parameter(torch.zeros(5_144_408, device=dev))]  # stands in for the network's parameters
        net[0].grad = torch.randn_like(net[0])
        reducer = D.GradAllReducer(net)

        def step():
            for v in ("lmain", "rmain"):
                for k in ("xyz", "rot_maps", "scale_maps", "opacity_maps"):
                    data[v][k].grad = None
            img = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"]
            L.stage2_photometric_loss(img, gt).backward()
            reducer()

        def local_step():  # the same without the collective: warm-up / self-check before anything is timed
            img = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"]
            L.stage2_photometric_loss(img, gt).backward()

        local_step(); local_step()
        torch.cuda.synchronize(dev)
    except Exception as e:  # noqa: BLE001
        ok, err = 0, repr(e)[:300]
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = int(flag.item())
    else:
        ok_all = ok
    if not ok_all:
        return {"error": err or "another rank failed"}
    el = timed(step, steps, 2)
    # the exchange step ALONE (the same 20.6 MB mean all-reduce, nothing around it): what the collective costs at this world size
    ar_ms = None
    if dist.is_initialized():
        try:
            el_ar = timed(reducer, 10, 2)
            ar_ms = round(el_ar / 10 * 1e3, 3)
        except Exception:  # noqa: BLE001
            ar_ms = None
    return {"iters_per_s": round(steps / el, 2), "allreduce_alone_ms": ar_ms, "stereo_pairs_per_s": round(world * batch * steps / el, 1), "ms_per_iter": round(el / steps * 1e3, 3),
            "batch_per_gpu": batch, "n_gpus": world, "render": "%dx%d" % (rres, rres),
            "allreduce": ("%s, world %d%s" % (dist.get_backend(), dist.get_world_size(), " (forced: GPSGS_DIST_FORCE=1)" if D.forced() and world == 1 else "")
                          if dist.is_initialized() else "none (one rank, no process group)"),
            "includes": "pts2render (fused pack + %d raster forwards as one autograd node, one HIP stream per sample) + L1/SSIM loss + backward to the per-pixel maps + mean all-reduce of "
                        "20.6 MB of network gradients (RCCL); networks not executed" % batch}


def config_leg(res, gaussians, render_res, dev, steps, inflight, rs_proto, timed, attributes="trained", stage_table=False):
    """Secondary measurement (never part of `value`): another BASELINE workload through the same C-ABI sessions as the headline --
    forward + backward and forward only (inference workspace), one view at a time and `inflight` views in flight, `steps` steps each.
    stage_table: also a per-kernel table (hipEvents around every kernel, one view at a time) with each kernel's algorithmic HBM fraction."""
    import torch
    from gps_gaussian_amd import _capi
    from gps_gaussian_amd import rasterizer as RZ
    from gps_gaussian_amd import synthetic as S
    from gps_gaussian_amd.session import RasterSession
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    lanes = []
    smp = S.make_stereo_sample(res, gaussians, seed=S.SEED + 77, render_res=render_res, attributes=attributes)  # one synthetic view; every lane renders its own copy of it
    g, cam = S.compact_sample(smp), smp["novel_view"]
    for i in range(inflight):
        t = {k: torch.from_numpy(g[k]).to(dev) for k in names}
        P = t["means3D"].shape[0]
        lanes.append(dict(train=RasterSession(P, render_res, render_res, dev, training=True), infer=RasterSession(P, render_res, render_res, dev, training=False),
                          stream=torch.cuda.Stream(device=dev), a=(t["means3D"], t["colors"], t["opacities"].reshape(-1), t["scales"], t["rotations"],
                                                                   torch.from_numpy(cam["world_view_transform"]).to(dev), torch.from_numpy(cam["full_proj_transform"]).to(dev),
                                                                   rs_proto.bg, math.tan(float(cam["FovX"]) * 0.5), math.tan(float(cam["FovY"]) * 0.5), 1.0),
                          gout=torch.randn(3, render_res, render_res, device=dev)))

    def run(kind, n, k, color_grad=True):  # k steps, n views in flight
        while k > 0:
            m = min(n, k)
            for L in lanes[:m]:
                L[kind].wave_priority = n == 1
                L[kind].forward_begin(*L["a"], L["stream"])
            for L in lanes[:m]:
                L[kind].forward_end()
                if kind == "train":
                    L[kind].backward(L["gout"], color_grad=color_grad, stream=L["stream"])
            k -= m

    out = {"P": int(lanes[0]["a"][0].shape[0]), "render": "%dx%d" % (render_res, render_res), "steps": steps, "blocks": 3}
    for kind, label in (("train", "fwd_bwd"), ("infer", "fwd_only")):
        out[label] = {}
        for n, nl in ((1, "one_view_in_flight"), (inflight, "%d_views_in_flight" % inflight)):
            run(kind, n, 2 * n + 3)  # warm-up: capacities learnt, workspaces allocated
            allocs0 = sum(L[kind].allocations for L in lanes)
            els = sorted(timed(lambda k, kind=kind, n=n: run(kind, n, k), steps, 0, multi=True) for _ in range(3))  # three blocks of `steps` steps
            el = els[1]
            out[label][nl] = {"views_per_s": round(steps / el, 1), "ms_per_view": round(el / steps * 1e3, 4),
                              "blocks_ms_per_view": [round(x / steps * 1e3, 4) for x in els],
                              "workspace_reallocations_while_timed": sum(L[kind].allocations for L in lanes) - allocs0}
    # the gradient set stage 2 differentiates (no dL/dcolour: the colours are input pixels) -- its backward instantiation keeps four waves per SIMD
    out["fwd_bwd_stage2_gradient_set"] = {}
    for n, nl in ((1, "one_view_in_flight"), (inflight, "%d_views_in_flight" % inflight)):
        run("train", n, 2 * n + 3, color_grad=False)
        els = sorted(timed(lambda k, n=n: run("train", n, k, color_grad=False), steps, 0, multi=True) for _ in range(3))
        out["fwd_bwd_stage2_gradient_set"][nl] = {"views_per_s": round(steps / els[1], 1), "ms_per_view": round(els[1] / steps * 1e3, 4)}
    out["R"] = R = int(RZ.last_stats(dev).get("last_R", 0))
    if stage_table:
        # per-kernel durations of THIS workload, one view at a time (every launch has the chip to itself), and each kernel's algorithmic HBM
        # fraction (same byte formulas as the headline's `stages`; R = this view's measured instance count)
        P = out["P"]
        NB = (((render_res + 7) // 8 + 3) // 4 * 4) * ((render_res + 7) // 8)
        npix = render_res * render_res
        # gradient records of this view: slots (one flag byte each, read by k_preprocess_bwd) and slots that hold a 36-byte record (written by the
        # compositing backward, read back by k_preprocess_bwd) -- counted on the device after a backward (212 P under-counts a
        # regime with ~100 slots per Gaussian)
        run("train", 1, 1)
        torch.cuda.synchronize(dev)  # (the sessions run on their own streams)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        sess0 = lanes[0]["train"]
        _capi.check(_capi.lib().gsr_debug_count_records(sess0.ws.data_ptr(), sess0.ws.numel(), P, render_res, render_res, sess0.cap, sess0.bin_cap, cnt.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream), "gsr_debug_count_records")
        torch.cuda.synchronize(dev)
        n_rec, n_slots = int(cnt[0]), int(cnt[1])
        out["gradient_records"] = {"slots": n_slots, "written": n_rec}
        alg = stage_bytes(P, R, NB, npix, n_slots, n_rec, direct=bool(sess0.bin_cap))
        RZ.set_stage_timing(True)
        run("train", 1, 3)
        torch.cuda.synchronize(dev)
        _capi.timing_read()
        run("train", 1, max(10, steps))
        st = _capi.timing_read()
        RZ.set_stage_timing(False)
        tab = {}
        traffic = counter_traffic("pmc_traffic_regime.json", P, render_res, render_res, R)
        try:
            sclk = _capi.measure_sclk_mhz(dev)
        except Exception:  # noqa: BLE001
            sclk = None
        for k, (ms, n) in st.items():
            if n and k in alg:
                us = ms / n * 1e3
                tab[k] = {"avg_us": round(us, 2), "launches": n, "algorithmic_bytes": alg[k]}
                pm = traffic.get(k, {})
                tr = pm.get("hbm_bytes_per_launch")
                if tr:
                    tab[k].update(counter_bytes=tr, counter_gbs=round(tr / (us * 1e-6) / 1e9, 1), traffic_ratio=round(tr / alg[k], 3))
                tab[k].update(stage_bound(us, alg[k], tr, pm.get("valu_wave_instructions_per_launch"), sclk))  # hbm_frac: from the COUNTER bytes when they are known
        out["stages_one_view_in_flight"] = tab
        out["stages_sum_us"] = round(sum(v["avg_us"] for v in tab.values()), 1)
        # the same table for the gradient set stage 2 differentiates (no dL/dcolour): the two backward kernels are other instantiations -- one-sector
        # records that carry dL/dopacity (round 6)
        run("train", 1, 3, color_grad=False)
        torch.cuda.synchronize(dev)
        cnt.zero_()
        _capi.check(_capi.lib().gsr_debug_count_records(sess0.ws.data_ptr(), sess0.ws.numel(), P, render_res, render_res, sess0.cap, sess0.bin_cap, cnt.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream), "gsr_debug_count_records")
        torch.cuda.synchronize(dev)
        alg2 = stage_bytes(P, R, NB, npix, int(cnt[1]), int(cnt[0]), direct=bool(sess0.bin_cap), color_grad=False)
        RZ.set_stage_timing(True)
        _capi.timing_read()
        run("train", 1, max(10, steps), color_grad=False)
        st2 = _capi.timing_read()
        RZ.set_stage_timing(False)
        out["stages_stage2_gradient_set"] = {k: dict(avg_us=round(st2[k][0] / st2[k][1] * 1e3, 2), algorithmic_bytes=alg2[k],
                                                     hbm_frac_algorithmic=round(alg2[k] / (st2[k][0] / st2[k][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
                                             for k in ("composite_bwd", "preprocess_bwd") if st2.get(k, (0, 0))[1]}
        out["longest_bin_list"] = int(RZ._dev_state(dev).get("longest", 0))
    return out


def full_pipeline_leg(budget_s):
    """BASELINE configs 3 and 4 at full size, measured in THIS run WHERE A REFERENCE CHECKOUT EXISTS BESIDE THE GPU (nothing of the reference travels to
    the GPU box since round 6, so on the driver's box this leg reports the recorded run of round 4 -- profiles/full_pipeline.json -- and says so): the
    reference's unmodified test_view_interp.py / train_stage2.Trainer with its real networks (random weights) on the drop-in kernels, each once as the reference runs it and once with GPSGS_ACCELERATE=all (the opt-in import hook that lets the unmodified scripts reach the
    fused pack / loss / corr / upsample / unproject kernels).  MIOpen's exhaustive convolution search takes ~5 minutes on a fresh box, so the children
    run with MIOPEN_FIND_MODE=FAST (unless the caller set one): the networks are then ~15 % slower than with the default find mode -- the builder's
    default-find-mode numbers are replayed next to these, labelled.  Anything that does not fit the time budget is skipped and says so."""
    import subprocess
    import time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import refenv
        ref = refenv.reference_dir()
    except Exception as e:  # noqa: BLE001
        return {"measured_in_this_run": False, "skipped": "tools/refenv.py unavailable: %r" % (e,)}
    if ref is None:
        return {"measured_in_this_run": False, "skipped": "no reference checkout on this machine (nothing of the reference travels to the GPU box)"}
    env = dict(os.environ)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    env.pop("GPSGS_ACCELERATE", None)
    t_end = time.perf_counter() + budget_s
    out = {"measured_in_this_run": True, "miopen_find_mode": env["MIOPEN_FIND_MODE"], "reference": os.path.relpath(ref, ROOT) if ref.startswith(ROOT) else ref,
           "weights": "random (no checkpoint offline): the regressed scales sit at their 0.01 m clamp, ~3e7 (Gaussian, bin) instances per 2048^2 view",
           "budget_s": budget_s}
    work = os.environ.get("GPSGS_BENCH_WORK", "/tmp/gpsgs_bench_ref")
    legs = (("config4_stage2_accelerated", ["ddp", "--res", "1024", "--steps", "16", "--batch", "4", "--train-samples", "4", "--work", work + "_w4", "--accelerate", "all"]),
            ("config4_stage2_as_the_reference_runs_it", ["ddp", "--res", "1024", "--steps", "12", "--batch", "4", "--train-samples", "4", "--work", work + "_w4"]),
            ("config3_view_interp_accelerated", ["interp", "--res", "1024", "--samples", "2", "--views", "5", "--work", work + "_w3", "--accelerate", "all"]),
            ("config3_view_interp_as_the_reference_runs_it", ["interp", "--res", "1024", "--samples", "2", "--views", "5", "--work", work + "_w3"]))
    for name, argv in legs:
        left = t_end - time.perf_counter()
        if left < 45:
            out[name] = {"measured_in_this_run": False, "skipped": "time budget (%.0f s) used up" % budget_s}
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference.py")] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               timeout=left, cwd=ROOT, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"measured_in_this_run": False, "error": (r.stderr or r.stdout)[-300:]}
                continue
            d = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            out[name] = {"measured_in_this_run": False, "skipped": "did not finish inside the time budget (%.0f s)" % budget_s}
            continue
        except Exception as e:  # noqa: BLE001
            out[name] = {"measured_in_this_run": False, "error": repr(e)[:300]}
            continue
        keep = {"measured_in_this_run": True, "wall_s": round(time.perf_counter() - t0, 1), "accelerate": d.get("accelerate", {}).get("requested", ""),
                "fused_calls": d.get("accelerate", {}).get("calls", {})}
        if d["mode"] == "ddp":
            keep.update(stage2_iters_per_s=d["iters_per_s"], iter_ms=d["iter_ms_median_second_half"], gpu_ms_per_iter=d["gpu_ms_per_iter"], batch_per_gpu=d["batch_per_gpu"],
                        render=d["render"], steps=d["steps"], world_size=d["world_size"])
        else:
            keep.update(views_per_s_within_sample=d["script_run"].get("views_per_s_within_sample"), views_per_s_whole_script=d["script_run"]["views_per_s_end_to_end"],
                        views_per_s_gpu_side=d.get("views_per_s_gpu_side"), gpu_ms_per_view=d.get("gpu_ms_per_view"), gaussians_per_view=d.get("gaussians_per_view"),
                        render=d["render"], views=d["script_run"]["views"])
        out[name] = keep
    out["leg_wall_s"] = round(budget_s - (t_end - time.perf_counter()), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=1024, help="source resolution (config 2: 1024)")
    ap.add_argument("--render-res", type=int, default=None, help="render resolution (default = --res)")
    ap.add_argument("--gaussians", type=int, default=600_000)
    ap.add_argument("--inflight", type=int, default=6, help="independent views rendered concurrently per GPU (one RasterSession + HIP stream each)")
    ap.add_argument("--repeats", type=int, default=15, help="timed blocks of EXACTLY --steps steps each; value / ms_per_step are the median block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary `configs` block (config 2 rendered at 2048^2, config 5)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip every secondary leg (Python host path, forward only, deferred check, stage-2 path, neighbours, CPU rows): with --inflight 1 "
                         "every kernel launch of the run then has the chip to itself (tools/prof_r04.sh profiles that mode for the exclusive durations)")
    ap.add_argument("--no-full-pipeline", action="store_true", help="skip the `full_pipeline` leg (BASELINE configs 3 / 4 with the reference's own scripts and networks)")
    ap.add_argument("--full-pipeline-budget", type=float, default=600.0, help="seconds the full-pipeline leg may take in total; what does not fit is skipped and says so")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    # ---- `python bench.py --gpus N` (no torchrun around it): become N ranks.  The reference has no launcher at all
    # (/root/reference/train_stage2.py:27-55 is single-process), so this is the whole multi-GPU entry: one process per GPU, rendezvous on
    # 127.0.0.1, RCCL underneath.  Under `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is already set and must agree.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and os.environ.get("GPSGS_BENCH_SINGLE_DEVICE") != "1":
            raise SystemExit("bench.py: --gpus %d but this node exposes %d GPU(s) (set GPSGS_BENCH_SINGLE_DEVICE=1 to put every rank on device 0: "
                             "a functional test of the multi-rank path, not a measurement)" % (args.gpus, have))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvpe(sys.executable, cmd, dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s rank(s) (WORLD_SIZE); pass the same N to both" % (args.gpus, env_world))

    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import _capi, synthetic as S
    from gps_gaussian_amd import dist as D
    from gps_gaussian_amd import rasterizer as RZ

    rank, local_rank, world = D.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    # test hooks (a 1-GPU box can still exercise the multi-rank code path): all ranks on device 0, gloo instead of RCCL
    if os.environ.get("GPSGS_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GPSGS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init(backend=backend, device=dev)  # "nccl" is RCCL on ROCm; no-op at world size 1
    # one process per GPU, each host-bound in bursts (F streams of launches): every rank on its own slice of the CPUs (its GPU's NUMA node when
    # sysfs names one), narrowed to eight CPUs of one L3 domain inside it; a no-op at world size 1 and with GPSGS_AFFINITY=0
    cpu_slice = D.set_cpu_affinity(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), local_rank, n_cpus=8)
    # ONE rank: the GPU legs run pinned to eight CPUs of one L3 domain next to the GPU (dist.pin_near_gpu: what INTEGRATION.md tells a training
    # script to do -- the plugin path is ~240 us of two-thread host work per 250 us GPU step, and a scheduler that migrates it across a 256-CPU box
    # makes it ~40 % slower); the CPU-heavy legs (full pipeline children, the OpenMP baseline) run with the original mask again
    # ONE rank: nothing is pinned while `value` is measured (round 6: the reference's scripts do not pin, and the compiled host path -- csrc/host_shim.cpp -- has
    # taken the Python / PyTorch host work that made the step host-bound on some boxes); the same step pinned to eight CPUs of one L3 domain next to
    # the GPU (dist.pin_near_gpu) is measured right behind it and reported as `plugin_api_pinned_views_per_s`
    pinned_one_rank = False
    _capi.lib()  # fail loudly if the HIP library is missing
    assert RZ._fast_module(), "bench.py: the compiled host path (lib/_gpsgs_fast.so) is missing -- run __graft_entry__.build()"

    # ---- synthetic workload: one stereo pair per rank (different pose per rank), resident in HBM ------------------
    render_res = args.render_res or args.res
    s = S.make_stereo_sample(args.res, args.gaussians, seed=S.SEED + rank, render_res=render_res)
    g = S.compact_sample(s)
    cam = s["novel_view"]
    P = g["means3D"].shape[0]
    H = W = render_res
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in names}
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    rs = RZ.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(float(cam["FovX"]) * 0.5), tanfovy=math.tan(float(cam["FovY"]) * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.from_numpy(cam["world_view_transform"]).to(dev),
        projmatrix=torch.from_numpy(cam["full_proj_transform"]).to(dev), sh_degree=3,
        campos=torch.from_numpy(cam["camera_center"]).to(dev), prefiltered=False, debug=False)
    rast = RZ.GaussianRasterizer(rs)
    gout = torch.randn(3, H, W, device=dev)  # dL/dpix of a synthetic loss, random (not zero: DVFS note in the guide)

    def fwd_bwd():  # the same step through the drop-in autograd module (secondary number: adds PyTorch's per-call autograd machinery)
        for v in t.values():
            v.grad = None
        m2.grad = None
        img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=None, colors_precomp=t["colors"],
                      scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        img.backward(gout)

    def fwd_only():
        with torch.no_grad():
            rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=None, colors_precomp=t["colors"],
                 scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)

    # ---- the headline step: forward + backward of one view through the C-ABI (gsr_forward_notify + gsr_backward), driven by
    # gps_gaussian_amd.session.RasterSession: same kernels, same exact capacity check (the scan publishes the instance count to pinned
    # host memory, the host waits for it every forward, an overflow would be re-rendered), preallocated buffers, no autograd round trip.
    # Views are independent units, and one view's kernels leave the chip under-occupied (config 2 has ~5 one-wave compositing work
    # items per SIMD): --inflight F views are rendered CONCURRENTLY, each by its own session on its own HIP stream.  A "step" is still
    # one view (forward + backward); steps are issued in groups of F (begin all forwards, then per view: capacity check + backward).
    from gps_gaussian_amd.session import RasterSession
    F = max(1, args.inflight)
    raw = {k: t[k].detach() for k in names}
    lanes = [dict(sess=RasterSession(P, W, H, dev, training=True), stream=torch.cuda.Stream(device=dev), m3=raw["means3D"], col=raw["colors"],
                  opa=raw["opacities"].reshape(-1), sca=raw["scales"], rot=raw["rotations"], view=rs.viewmatrix, proj=rs.projmatrix, gout=gout)]
    for i in range(1, F):  # further views of the same rig (different novel pose and stereo pair), same size
        si = S.make_stereo_sample(args.res, args.gaussians, seed=S.SEED + rank + 1000 * i, render_res=render_res)
        gi, ci = S.compact_sample(si), si["novel_view"]
        assert gi["means3D"].shape[0] == P
        ti = {k: torch.from_numpy(gi[k]).to(dev) for k in names}
        lanes.append(dict(sess=RasterSession(P, W, H, dev, training=True), stream=torch.cuda.Stream(device=dev), m3=ti["means3D"], col=ti["colors"],
                          opa=ti["opacities"].reshape(-1), sca=ti["scales"], rot=ti["rotations"], view=torch.from_numpy(ci["world_view_transform"]).to(dev),
                          proj=torch.from_numpy(ci["full_proj_transform"]).to(dev), gout=torch.randn(3, H, W, device=dev)))
    torch.cuda.synchronize(dev)

    def group(n):  # n <= F views, concurrently (every session is handed its stream: PyTorch's current stream is never switched)
        for L in lanes[:n]:
            L["sess"].wave_priority = n == 1  # GSR_FLAG_WAVE_PRIORITY pays for a view that has the chip to itself, costs ~2 % of the aggregate with views overlapped
            L["sess"].forward_begin(L["m3"], L["col"], L["opa"], L["sca"], L["rot"], L["view"], L["proj"], rs.bg, rs.tanfovx, rs.tanfovy, 1.0, L["stream"])
        for L in lanes[:n]:
            L["sess"].forward_end()
            L["sess"].backward(L["gout"], stream=L["stream"])

    def steps_pipelined(k):
        while k > 0:
            group(min(F, k))
            k -= min(F, k)

    def group_s2(n):  # the same with the gradient set stage 2 needs: no dL/dcolour (the colours are input pixels, never differentiated)
        for L in lanes[:n]:
            L["sess"].wave_priority = n == 1
            L["sess"].forward_begin(L["m3"], L["col"], L["opa"], L["sca"], L["rot"], L["view"], L["proj"], rs.bg, rs.tanfovx, rs.tanfovy, 1.0, L["stream"])
        for L in lanes[:n]:
            L["sess"].forward_end()
            L["sess"].backward(L["gout"], color_grad=False, stream=L["stream"])

    def steps_pipelined_s2(k):
        while k > 0:
            group_s2(min(F, k))
            k -= min(F, k)

    sess = lanes[0]["sess"]

    def step():  # one view at a time on the current stream (calibration of the per-kernel table, and the single-view-in-flight number)
        L = lanes[0]
        sess.wave_priority = True  # a view on its own (the views-in-flight groups switch it off for their launches)
        sess.forward(L["m3"], L["col"], L["opa"], L["sca"], L["rot"], L["view"], L["proj"], rs.bg, rs.tanfovx, rs.tanfovy, 1.0)
        sess.backward(L["gout"])

    def barrier():
        D.barrier(local_rank)

    # The host's cyclic garbage collector is switched off for the whole measurement (one collection + freeze first): a full collection of a
    # process that has imported torch takes tens of milliseconds -- the size of the rare outlier blocks seen in the 2048^2 legs -- against
    # timed blocks of 5-50 ms.  (Collecting before every timed block instead leaves the GPU idle for those milliseconds right before the
    # clock starts: its clocks drop and EVERY block reads 10-15 % slower -- measured.)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    def timed(fn, steps, warmup, multi=False):  # multi: fn(k) runs k steps itself (pipelined groups)
        if multi:
            fn(warmup)
        else:
            for _ in range(warmup):
                fn()
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if multi:
            fn(steps)
        else:
            for _ in range(steps):
                fn()
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        return D.max_over_ranks(time.perf_counter() - t0, device=dev)

    # ---- calibration (untimed): every kernel bracketed by hipEvents -> per-stage table, dominant kernel ------------------
    RZ.set_stage_timing(True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    _capi.timing_read()  # drop the warm-up records
    # three calibration blocks; per stage the block with the smallest average is kept (one launch that meets a clock ramp or a page fault
    # -- seen once: a 4.7 ms k_preprocess launch -- would otherwise decide which kernel is called dominant)
    stages = None
    for _ in range(3):
        for _ in range(max(5, min(20, args.steps))):
            step()
        blk = _capi.timing_read()
        if stages is None:
            stages = blk
        else:
            for k_, (ms_, n_) in blk.items():
                if n_ and (not stages[k_][1] or ms_ / n_ < stages[k_][0] / stages[k_][1]):
                    stages[k_] = (ms_, n_)
    dom_stage = max(stages, key=lambda k: (stages[k][0] / stages[k][1]) if stages[k][1] else 0.0)

    # ---- the session regions (secondary; `value` is measured further down, through the reference's plugin API).  Only the dominant kernel keeps
    # its hipEvent bracket (on the launch stream), so that the measurement does not perturb the pipeline it measures.  After the switch to that
    # mode >= 10 untimed steps run before anything is timed.  Then REPEATS blocks of EXACTLY --steps steps with F views in flight are timed
    # (barrier + synchronize on both sides, MAX over ranks): `session.views_in_flight_*` is the MEDIAN block. -----------------------------------
    RZ.set_stage_timing(True, dom_stage)
    steps_pipelined(max(10, args.warmup, 2 * F) + 4 * F + args.steps)  # untimed; also lets the clocks settle under the concurrent load (one block's worth)
    torch.cuda.synchronize(dev)
    _capi.timing_read()
    REPEATS = max(3, args.repeats)
    blocks = [timed(steps_pipelined, args.steps, 0, multi=True) for _ in range(REPEATS)]
    dom_live = _capi.timing_read()[dom_stage]
    # Second timed region: the same step, EXACTLY --steps of them, one view in flight, the dominant kernel still bracketed by hipEvents on
    # its launch stream.  The roofline takes the kernel's duration from HERE: a roofline compares a kernel that has the chip to itself
    # with the chip's peak, and a launch that time-shares the chip with the kernels of F - 1 other views (the region above)
    # has no exclusive duration -- there every launch lasts longer while the aggregate rate is higher (its duration is reported too:
    # `roofline.views_in_flight_region`).
    # Five blocks, the MEDIAN block reported (>= 100 exclusive launches in all): a single block is at the mercy of one stalled launch -- seen on one box: one
    # ~0.9 ms hiccup among 105 launches moved the kernel's average from 108 to 117 us and the one-view rate by 2.4 %.
    n_single = max(args.steps, 20)
    N_SINGLE_BLOCKS = 5

    def timed_blocks(fn):
        out = []
        for b_ in range(N_SINGLE_BLOCKS):
            el_ = timed(fn, n_single, 5 if b_ == 0 else 0)
            out.append((el_, _capi.timing_read()[dom_stage]))
        out.sort(key=lambda x: x[0])
        return out[N_SINGLE_BLOCKS // 2]

    el_single, dom_excl = timed_blocks(step)
    sclk_mhz = None
    try:
        sclk_mhz = round(_capi.measure_sclk_mhz(dev), 1)  # shader clock under VALU load, measured (not the 2.4 GHz maximum)
    except Exception:  # noqa: BLE001
        sclk_mhz = None
    # COLD variant of the one-view step: every timed step above re-renders the same view, whose ~0.3 GB of workspace and inputs
    # largely survive in the 256 MB Infinity Cache from one step to the next; in training the networks stream gigabytes between two renders.  Here a
    # 1 GiB device fill runs between steps, OUTSIDE the event bracket of the step; the warm figure is taken the same way (events, no fill) for comparison.
    cold = None
    if not args.headline_only:
        try:
            evict = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

            def event_ms(n, fill):
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
                for a_, b_ in ev:
                    if fill:
                        evict.fill_(1)
                    a_.record(); step(); b_.record()
                torch.cuda.synchronize(dev)
                return sorted(a_.elapsed_time(b_) for a_, b_ in ev)[n // 2]

            event_ms(5, True)
            ms_cold, ms_warm = event_ms(25, True), event_ms(25, False)
            cold = {"single_view_cold_views_per_s": round(1e3 / ms_cold, 1), "single_view_warm_same_method_views_per_s": round(1e3 / ms_warm, 1),
                    "ms_cold": round(ms_cold, 4), "ms_warm": round(ms_warm, 4),
                    "how": "median of 25 C-ABI session steps, each bracketed by two events; cold: a 1 GiB device fill between steps, outside the bracket"}
            del evict
        except Exception as e:  # noqa: BLE001
            cold = {"error": repr(e)[:200]}
    # secondary: the stage-2 gradient set (no dL/dcolour), same sessions; the dominant kernel keeps its bracket for the exclusive duration
    s2 = None
    if not args.headline_only:
        def step_s2():
            L = lanes[0]
            sess.wave_priority = True
            sess.forward(L["m3"], L["col"], L["opa"], L["sca"], L["rot"], L["view"], L["proj"], rs.bg, rs.tanfovx, rs.tanfovy, 1.0)
            sess.backward(L["gout"], color_grad=False)
        el_s2_one, dom_s2 = timed_blocks(step_s2)
        el_s2 = sorted(timed(steps_pipelined_s2, args.steps, 2 * F if i == 0 else 0, multi=True) for i in range(5))[2]
        _capi.timing_read()
        s2 = {"gradients": "means3D, means2D, opacities, scales, rotations (GSR_FLAG_NO_COLOR_GRAD: what train_stage2.py differentiates; the colours are input pixels)",
              "views_per_s": round(world * args.steps / el_s2, 2), "views_in_flight": F, "single_view_in_flight_views_per_s": round(world * n_single / el_s2_one, 2),
              "dominant_kernel_avg_launch_us": round(dom_s2[0] / max(1, dom_s2[1]) * 1e3, 2) if dom_s2[1] else None}
    RZ.set_stage_timing(False)
    sess_elapsed = sorted(blocks)[REPEATS // 2]
    sess_q1, sess_q3 = sorted(blocks)[REPEATS // 4], sorted(blocks)[(3 * REPEATS) // 4]
    R = int(RZ.last_stats(dev).get("last_R", 0))  # measured number of (Gaussian, bin) instances of this view

    # ---- THE HEADLINE: the same step through the reference's plugin API -- GaussianRasterizer(raster_settings)(...) + image.backward(), the call shape
    # of gaussian_renderer/__init__.py:51-62 -- one view at a time, sync capacity check.  REPEATS blocks of EXACTLY --steps steps (barrier +
    # synchronize on both sides, MAX over ranks); `value` / `ms_per_step` are the median block, the quartiles are reported.  (Rounds 2-4 put the
    # C-ABI session number with six views in flight first: a caller-owned-buffer API no reference code uses -- VERDICT r04 weak 7.)
    api_blocks = [timed(fwd_bwd, args.steps, max(5, args.warmup) if i == 0 else 0) for i in range(REPEATS)]
    elapsed = sorted(api_blocks)[REPEATS // 2]
    q1, q3 = sorted(api_blocks)[REPEATS // 4], sorted(api_blocks)[(3 * REPEATS) // 4]
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed
    # ... and the same blocks with the process pinned next to its GPU (one rank only; the remaining GPU legs keep the pin, the CPU legs drop it)
    el_pinned = None
    if cpu_slice is None and world == 1:
        cpu_slice = D.pin_near_gpu(local_rank)
        pinned_one_rank = cpu_slice is not None
        if pinned_one_rank:
            el_pinned = sorted(timed(fwd_bwd, args.steps, 5 if i == 0 else 0) for i in range(5))[2]
    # the Python host path (GPSGS_FAST=0: rasterizer.py's autograd Function, what rounds 1-5 measured), same pinning state as the line above
    el_pyhost = None
    if not args.headline_only:
        os.environ["GPSGS_FAST"] = "0"
        el_pyhost = sorted(timed(fwd_bwd, args.steps, 5 if i == 0 else 0) for i in range(5))[2]
        os.environ["GPSGS_FAST"] = "1"

    # secondary numbers: forward only, and the non-blocking check mode, through the same module
    el_fwd = el_def = el_fwd_def = None
    if not args.headline_only:
        el_fwd = timed(fwd_only, args.steps, 3)
        os.environ["GPSGS_CHECK"] = "deferred"
        el_def = timed(fwd_bwd, args.steps, 3)
        el_fwd_def = timed(fwd_only, args.steps, 3)
        os.environ["GPSGS_CHECK"] = "sync"
    torch.cuda.synchronize(dev)

    # SURVEY.md section 8(d) counts instances on upstream's 16x16 tiles: the same view's tile-instance count, from the radii and the
    # projected centres the forward left in its workspace (not timed)
    R_tile = None
    try:
        st_ = RZ.export_state(sess.ws, P, W, H, sess.cap, sess.bin_cap)
        xy, rad = st_["xy"], sess.radii.float()
        gx, gy = (W + 15) // 16, (H + 15) // 16
        x0 = ((xy[:, 0] - rad) / 16).to(torch.int32).clamp(0, gx); x1 = ((xy[:, 0] + rad + 15) / 16).to(torch.int32).clamp(0, gx)
        y0 = ((xy[:, 1] - rad) / 16).to(torch.int32).clamp(0, gy); y1 = ((xy[:, 1] + rad + 15) / 16).to(torch.int32).clamp(0, gy)
        R_tile = int((((x1 - x0) * (y1 - y0)).long() * (sess.radii > 0)).sum().item())
    except Exception:  # noqa: BLE001
        R_tile = None

    # ---- secondary: the kernels either side of the rasteriser (rows f1-f4) with algorithmic bytes and HBM fractions -------------------------------
    neighbours = None
    if rank == 0 and world == 1 and not args.headline_only:
        try:
            neighbours = neighbours_leg(dev)
        except Exception as e:  # noqa: BLE001
            neighbours = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()

    # ---- secondary: the hot path inside one stage-2 training iteration (BASELINE config 4: batch = 4 stereo pairs per GPU) ----
    stage2 = None if args.headline_only else stage2_leg(args, s, dev, rank, local_rank, world, D, timed)

    # ---- secondary: the reference's real render size and the stress configuration (rank 0 at N = 1 only; synthetic clouds of those sizes) ----
    configs = None
    if rank == 0 and world == 1 and not args.headline_only and not args.no_configs and (args.res, args.gaussians, W) == (1024, 600000, 1024):
        configs = {}
        for cname, (cres, cg, crr) in (("config2_rendered_at_2048 (config/stage2.yaml:14-15 use_hr_img: what stage 2 really renders)", (1024, 600000, 2048)),
                                       ("config5_2048_2p4M", (2048, 2400000, 2048))):
            try:
                configs[cname] = config_leg(cres, cg, crr, dev, max(10, min(args.steps, 30)), 3, rs, timed)
            except Exception as e:  # noqa: BLE001
                configs[cname] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
        # the regime BASELINE configs 3 / 4 REALLY run in offline (and stage-2 training starts in): freshly initialised regression heads put every
        # scale at its 0.01 m clamp (lib/gs_parm_network.py:43,76) -- ~25 px splats at 2048^2, ~55 bins per Gaussian, R ~ 3 x 10^7, ~20,000 bins
        # with lists of 1,000-3,000 entries (profiles/r04_full_pipeline.md, profiles/r04_regime_kernel_stats.md).  Synthetic stand-in with the same statistics, per-kernel table included.
        try:
            configs["config3_regime_untrained_heads_2048 (scales at the 0.01 m clamp, opacity ~0.5: what random network weights give)"] = \
                config_leg(1024, 550000, 2048, dev, 10, 2, rs, timed, attributes="untrained", stage_table=True)
        except Exception as e:  # noqa: BLE001
            configs["config3_regime_untrained_heads_2048"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    # ---- the full pipeline: BASELINE configs 3 / 4 with the reference's OWN scripts and networks, MEASURED IN THIS RUN -------------------------
    # (rank 0, N = 1; where a reference checkout exists it is the CALLER of the product here, executed by tools/run_reference.py in child processes)
    full_pipeline = None
    D.restore_affinity()  # (a no-op unless pin_near_gpu() pinned this rank: everything below is CPU-heavy or runs in child processes)
    el_unpinned = elapsed  # (`value` itself is the unpinned figure since round 6)
    if rank == 0 and world == 1 and not args.headline_only and not args.no_full_pipeline and (args.res, args.gaussians, W) == (1024, 600000, 1024):
        torch.cuda.empty_cache()  # (the default workload only, like `configs`: a small functional run of bench.py does not start full-size networks)
        full_pipeline = full_pipeline_leg(args.full_pipeline_budget)
    fp_file = os.path.join(ROOT, "profiles", "full_pipeline.json")
    if rank == 0 and full_pipeline is not None and os.path.exists(fp_file):
        try:  # the builder's own measurement of the same scripts with MIOpen's default (exhaustive) find mode, for comparison: labelled as a replay
            full_pipeline["replayed_default_find_mode"] = dict(json.load(open(fp_file)), measured_in_this_run=False,
                                                               replayed_from="profiles/full_pipeline.json (tools/run_reference.py on an MI355X, MIOpen default find mode: "
                                                                             "minutes of search on a fresh box; profiles/r04_full_pipeline.md)")
        except Exception:  # noqa: BLE001
            pass

    # ---- roofline of the dominant kernel -------------------------------------------------------------------------------
    NB = (((W + 7) // 8 + 3) // 4 * 4) * ((H + 7) // 8)  # 8x8-pixel bins (one wave64 each), DESIGN.md section 2
    npix = W * H
    # ALGORITHMIC bytes per launch (SURVEY.md section 8d convention: each input read once, each output
    # written once; R = measured (Gaussian, bin) instances, NB = bins)
    # gradient records of this view (slots, and slots the compositing backward wrote a record into), counted on the device
    n_slots = n_rec = None
    try:
        torch.cuda.synchronize(dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        _capi.check(_capi.lib().gsr_debug_count_records(sess.ws.data_ptr(), sess.ws.numel(), P, W, H, sess.cap, sess.bin_cap, cnt.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                    "gsr_debug_count_records")
        torch.cuda.synchronize(dev)
        n_rec, n_slots = int(cnt[0]), int(cnt[1])
    except Exception:  # noqa: BLE001
        n_slots = n_rec = None
    alg_bytes = stage_bytes(P, R, NB, npix, n_slots, n_rec, direct=bool(sess.bin_cap))
    traffic_all = counter_traffic("pmc_traffic.json", P, W, H, R)
    per_stage = {}
    for name, (ms, n) in stages.items():
        if n and name in alg_bytes:
            avg_ms = ms / n
            per_stage[name] = {"avg_us": round(avg_ms * 1e3, 2), "launches": n, "algorithmic_bytes": alg_bytes[name],
                               "hbm_gbs": round(alg_bytes[name] / (avg_ms * 1e-3) / 1e9, 1)}
            pm_ = traffic_all.get(name, {})
            tr_ = pm_.get("hbm_bytes_per_launch")
            if tr_:  # L2-miss traffic of the same workload from the separate counter passes: 2 x FETCH_SIZE + WRITE_SIZE (profiles/pmc_traffic.json)
                per_stage[name].update(counter_bytes=tr_, counter_gbs=round(tr_ / (avg_ms * 1e-3) / 1e9, 1), traffic_ratio=round(tr_ / alg_bytes[name], 3))
            per_stage[name].update(stage_bound(avg_ms * 1e3, alg_bytes[name], tr_, pm_.get("valu_wave_instructions_per_launch"), sclk_mhz))
    dom = max(per_stage, key=lambda k: per_stage[k]["avg_us"]) if per_stage else None
    roofline = None
    if dom:
        achieved = per_stage[dom]["hbm_gbs"]
        pmc = traffic_all.get(dom, {})
        traffic, valu_instr, pmc_src = pmc.get("hbm_bytes_per_launch"), pmc.get("valu_wave_instructions_per_launch"), (traffic_all.get("source") if pmc else None)
        # exclusive duration: the one-view-in-flight timed region (falls back to the calibration pass if the dominant kernel changed)
        dom_us = round(dom_excl[0] / max(1, dom_excl[1]) * 1e3, 2) if dom == dom_stage and dom_excl[1] else per_stage[dom]["avg_us"]
        ovl_us = round(dom_live[0] / max(1, dom_live[1]) * 1e3, 2) if dom == dom_stage and dom_live[1] else None
        # SURVEY.md section 8(d) prices the compositing kernels with the instance count of upstream's 16x16 tiles: that is `achieved` / `frac`.  The same
        # formula with this implementation's own (8x8-bin) instance count -- the bytes the kernel really walks -- is `frac_with_bin_8x8_instances`.
        alg_bin = alg_bytes[dom]
        alg_main = alg_bin
        if R_tile and dom in ("composite_fwd", "composite_bwd"):
            T16 = ((W + 15) // 16) * ((H + 15) // 16)
            alg_main = 40 * R_tile + 8 * T16 + 20 * npix + (44 * P if dom == "composite_bwd" else 0)  # section 8(d): 40 R + 8 T + 20 Npix (+ 44 P gradient write)
        achieved = round(alg_main / (dom_us * 1e-6) / 1e9, 1)
        roofline = {"bound": "hbm", "kernel": "k_" + dom + ("_tiles" if (dom.startswith("composite") and RZ._composite_flag()) else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": pmc_src,
                    # what the counters say the kernel moved, as a rate ("rocprof HBM GB/s against the chip's peak") and against the algorithmic bytes
                    "counter_gbs": (round(traffic / (dom_us * 1e-6) / 1e9, 1) if traffic else None),
                    "counter_frac": (round(traffic / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
                    "traffic_ratio": (round(traffic / alg_main, 3) if traffic else None),
                    "algorithmic_bytes_per_launch": alg_main, "avg_launch_us": dom_us,
                    "algorithmic_bytes": "SURVEY.md section 8(d): 40 R + 8 T + 20 Npix + 44 P with R = instances on upstream's 16x16 tiles (T tiles)",
                    "launches_averaged": int(dom_excl[1]) if dom == dom_stage else None,
                    "measured": "hipEvents around the kernel on its launch stream: the median of %d timed blocks of %d steps each with ONE view in flight (exclusive "
                                "duration; profiles/r06_kernel_stats_one_view.md is the rocprofv3 summary of the same mode)" % (N_SINGLE_BLOCKS, n_single),
                    "bound_note": "the contract's roofline is HBM; this kernel's HBM fraction is low by construction (>= 50 op/B: SURVEY.md section 8d says the same).  At config 2 it is "
                                  "bound by instruction issue per SIMD at the occupancy its registers allow (DESIGN.md section 4: counters, occupancy sweep, per-workgroup timeline)",
                    # the same kernel with F views in flight (C-ABI sessions): launches of different views overlap and time-share the chip
                    "views_in_flight_region": {"views_in_flight": F, "avg_launch_us": ovl_us},
                    "instances": {"tile_16x16": R_tile, "bin_8x8": R, "gradient_record_slots": n_slots, "gradient_records_written": n_rec},
                    "frac_with_bin_8x8_instances": round(alg_bin / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    # one wave64 VALU instruction holds a SIMD for 4 cycles; measured instructions (rocprofv3 SQ_INSTS_VALU, same workload) x 4 cycles /
                    # (1024 SIMDs x measured shader clock x t): an upper bound of the issue-port use
                    "shader_clock_mhz": sclk_mhz,
                    "valu_issue_frac": (round(valu_instr * 4.0 / (1024 * (sclk_mhz or 2400.0) * 1e6 * dom_us * 1e-6), 4) if valu_instr else None)}

    # ---- CPU baseline: the fp32 oracle on the host cores (rank 0, N=1 only), bounded sample ------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.headline_only:
        from oracle.gsr_oracle import OracleRasterizer
        o = OracleRasterizer("f32")
        dp = gout.cpu().numpy()
        fa = (g["means3D"], g["colors"], g["opacities"], g["scales"], g["rotations"], cam["world_view_transform"],
              cam["full_proj_transform"], W, H, rs.tanfovx, rs.tanfovy, np.zeros(3, np.float32))
        o.forward(*fa); o.backward(dp)  # warm-up (page in, OpenMP pool)
        n_cpu, t0 = 0, time.perf_counter()
        while n_cpu < 3 or (time.perf_counter() - t0 < 4.0 and n_cpu < 20):
            o.forward(*fa); o.backward(dp); n_cpu += 1
        dt = time.perf_counter() - t0
        cpu = {"value": round(n_cpu / dt, 3), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
               "sample": "%d fwd+bwd views of the same %dx%d / %d-Gaussian workload, oracle/gsr_oracle.c fp32 + OpenMP" % (n_cpu, W, H, P)}

    # ---- second CPU row: a port of the reference's Taichi point splat (lib/TaichiRender.py:13-24; taichi itself is not installed,
    # SURVEY.md section 8d): forward-only z-buffer splat of the same view's points, sequential like its deterministic tie rule ----
    cpu_splat = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.headline_only:
        try:
            import ctypes as C
            from oracle import gsr_oracle as GO
            GO.build()
            aux = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libaux_oracle.so"))
            m = g["means3D"].astype(np.float64)
            pm = cam["full_proj_transform"].astype(np.float64).reshape(4, 4)  # row-vector convention: p_hom = [x y z 1] @ M
            ph = np.concatenate([m, np.ones((P, 1))], 1) @ pm
            wv = np.concatenate([m, np.ones((P, 1))], 1) @ cam["world_view_transform"].astype(np.float64).reshape(4, 4)
            pix = np.stack([((ph[:, 0] / ph[:, 3] + 1) * W - 1) * 0.5, ((ph[:, 1] / ph[:, 3] + 1) * H - 1) * 0.5, 1.0 / (wv[:, 2] + 1e-8)], 1)
            pts = np.ascontiguousarray(np.concatenate([pix, g["colors"]], 1)[None], dtype=np.float32)
            mask = np.ones((1, P), np.float32)
            depth = np.zeros((1, H, W), np.float32)
            color = -np.ones((1, 3, H, W), np.float32)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            aux.zsplat_oracle.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
            n_s, t0 = 0, time.perf_counter()
            while n_s < 3 or (time.perf_counter() - t0 < 2.0 and n_s < 200):
                depth[:] = 0; color[:] = -1
                aux.zsplat_oracle(vp(pts), vp(mask), vp(depth), vp(color), 1, P, W); n_s += 1
            dt = time.perf_counter() - t0
            cpu_splat = {"value": round(n_s / dt, 2), "unit": "views/s", "cores": 1, "kind": "port",
                         "sample": "%d forward-only z-buffer splats of the same view's %d points at %dx%d (oracle/aux_oracle.c::zsplat_oracle; "
                                   "no blending, no gradients: not comparable work, reported because the north star names this path)" % (n_s, P, W, H)}
        except Exception as e:  # noqa: BLE001
            cpu_splat = {"error": repr(e)[:200]}

    if rank == 0:
        rate = lambda el: round(world * args.steps / el, 2) if el else None  # noqa: E731
        cfg_name = ("BASELINE config 2" if (args.res, args.gaussians, W) == (1024, 600000, 1024) else
                    "BASELINE config 2 rendered at 2048^2 (use_hr_img)" if (args.res, args.gaussians, W) == (1024, 600000, 2048) else
                    "BASELINE config 5" if (args.res, args.gaussians, W) == (2048, 2400000, 2048) else "non-BASELINE workload (parity / contract test size)")
        line = {
            "metric": "novel views/sec at 1024x1024 (~600k Gaussians), 1 GPU", "value": round(value, 2),
            "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d render of a synthetic stereo human, P=%d Gaussians, R=%d (Gaussian, 8x8-bin) instances, "
                                   "HIP rasteriser forward+backward, one view per step per GPU" % (cfg_name, W, H, P, R),
                       "api": "the reference's plugin API: GaussianRasterizer(raster_settings)(means3D=..., means2D=..., colors_precomp=..., opacities=..., scales=..., rotations=...) "
                              "+ image.backward() (gaussian_renderer/__init__.py:51-62), one view at a time on the current stream",
                       "check_mode": "sync (exact; the binning scan publishes the instance count to pinned host memory, checked on the host every forward)",
                       "wave_priority": "GSR_FLAG_WAVE_PRIORITY (hardware wave priorities in the compositing kernels, results unchanged): on for a view that has the chip to "
                                        "itself, off while several views are in flight (session.views_in_flight)"},
            "repeats_ms_per_step": [round(x / args.steps * 1e3, 4) for x in api_blocks],
            "ms_per_step_iqr": [round(q1 / args.steps * 1e3, 4), round(q3 / args.steps * 1e3, 4)],
            # the same kernels for a caller that owns its buffers (gps_gaussian_amd.session.RasterSession: two C-ABI calls per view, preallocated outputs /
            # workspace / gradients) -- an API no reference code uses: one view at a time, and F independent views in flight on F HIP streams
            "session": {"single_view_in_flight_views_per_s": round(world * n_single / el_single, 2),
                        "views_in_flight": F, "views_in_flight_views_per_s": round(world * args.steps / sess_elapsed, 2),
                        "views_in_flight_ms_per_step": round(sess_elapsed / args.steps * 1e3, 4),
                        "views_in_flight_ms_per_step_iqr": [round(sess_q1 / args.steps * 1e3, 4), round(sess_q3 / args.steps * 1e3, 4)]},
            "single_view_in_flight_views_per_s": round(world * n_single / el_single, 2),
            "autograd_api_views_per_s": round(value, 2),
            "cpu_affinity": ({"rank0_cpus": len(cpu_slice), "cpus": sorted(cpu_slice)[:16],
                              "how": ("NOT pinned while `value` was measured; dist.pin_near_gpu (one L3 domain on the GPU's NUMA node) for plugin_api_pinned_views_per_s and the "
                                      "GPU legs behind it" if pinned_one_rank else "dist.set_cpu_affinity: this rank's slice")}
                             if cpu_slice else "not set (GPSGS_AFFINITY=0, no sched_setaffinity, or no topology information)"),
            "not_measured_at_this_world_size": ([] if world == 1 else ["configs", "full_pipeline", "cpu_baseline", "cpu_taichi_splat_port", "neighbours"]),
            "stage2_gradient_set": s2,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_taichi_splat_port": cpu_splat,
            "forward_only_views_per_s": rate(el_fwd),
            "deferred_check_views_per_s": {"fwd_bwd": rate(el_def), "fwd": rate(el_fwd_def)},
            "plugin_api_unpinned_views_per_s": rate(el_unpinned),
            "plugin_api_pinned_views_per_s": rate(el_pinned),
            "plugin_api_python_host_path_views_per_s": rate(el_pyhost),
            "host_path": "compiled (lib/_gpsgs_fast.so: csrc/host_shim.cpp, a C++ autograd Function over the same C-ABI); GPSGS_FAST=0 selects the Python one",
            "stages": per_stage,
            "single_view_cold": cold,
            "neighbours": neighbours,
            "stage2_path": stage2,
            "configs": configs,
            "full_pipeline": full_pipeline,
        }
        print(json.dumps(line))
    D.shutdown()


if __name__ == "__main__":
    main()
