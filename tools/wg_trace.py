#!/usr/bin/env python
"""Per-SIMD timeline of the tile compositing kernels (development aid; MI355X).

    python tools/wg_trace.py [--res 1024] [--gaussians 600000] [--render-res R] [--out gpurun_out/wg_trace.npz] [--hint H]

Renders one forward + backward of the bench workload with gsr_debug_set_wg_trace on: every compositing workgroup (= one wave64 = one 8x8
bin) records when it started and ended (100 MHz wall clock), the shader cycles it spent, which XCD / SE / CU / SIMD it ran on and how
long its list was.  Printed per kernel: the span of the launch, the work per SIMD (sum of list lengths: the static balance), the
busy time per SIMD (union of its waves' intervals), how many waves a SIMD ran and how many at once, when the SIMDs ran dry (the tail),
and the order in which the dispatcher handed the first workgroups to the SIMDs.  --hint overrides the work-order hint (0 = plain patch order).
"""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(rows, name, sclk_mhz):
    ok = rows[:, 1] > 0
    r = rows[ok]
    if not len(r):
        return {"kernel": name, "workgroups": 0}
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0  # us
    cyc = r[:, 2].astype(np.float64)
    hw = r[:, 3] & 0xffffffff
    xcc = (r[:, 3] >> 32) & 0xff
    length = (r[:, 3] >> 40).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
    uniq, inv = np.unique(key, return_inverse=True)
    n = len(uniq)
    work = np.bincount(inv, weights=length, minlength=n)
    waves = np.bincount(inv, minlength=n)
    busy, last_end, peak = np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
    for s in range(n):
        m = inv == s
        iv = sorted(zip(start[m], end[m]))
        cur_s, cur_e, tot = iv[0][0], iv[0][1], 0.0
        for a, b in iv[1:]:
            if a > cur_e:
                tot += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy[s] = tot + cur_e - cur_s
        last_end[s] = max(b for _, b in iv)
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        c = p = 0
        for _, d in ev:
            c += d
            p = max(p, c)
        peak[s] = p
    span = float(end.max())
    order = np.argsort(r[:, 0], kind="stable")
    first = key[order][: 4 * n]
    # how the dispatcher fills the chip: do the first n workgroups land on n different SIMDs (breadth first)?
    breadth = len(np.unique(first[:n])) / n
    q = lambda a, p: float(np.quantile(a, p))  # noqa: E731
    return {
        "kernel": name, "workgroups": int(len(r)), "simds_used": int(n), "span_us": round(span, 2),
        "list_length": {"mean": round(float(length.mean()), 1), "max": int(length.max())},
        "wave_duration_us": {"mean": round(float((end - start).mean()), 2), "max": round(float((end - start).max()), 2)},
        "cycles_per_list_entry": round(float(cyc.sum() / max(1, length.sum())), 1),
        "work_per_simd (sum of list lengths)": {"mean": round(float(work.mean()), 1), "max": int(work.max()), "max_over_mean": round(float(work.max() / work.mean()), 3),
                                                "q05": q(work, 0.05), "q95": q(work, 0.95)},
        "waves_per_simd": {"mean": round(float(waves.mean()), 2), "min": int(waves.min()), "max": int(waves.max())},
        "peak_concurrent_waves_per_simd": {"mean": round(float(peak.mean()), 2), "max": int(peak.max())},
        "busy_us_per_simd": {"mean": round(float(busy.mean()), 2), "min": round(float(busy.min()), 2), "max": round(float(busy.max()), 2)},
        "simd_utilisation (mean busy / span)": round(float(busy.mean() / span), 3),
        "simd_ran_dry_at_us": {"q05": round(q(last_end, 0.05), 2), "q25": round(q(last_end, 0.25), 2), "median": round(q(last_end, 0.5), 2),
                               "q75": round(q(last_end, 0.75), 2), "max": round(float(last_end.max()), 2)},
        "first_n_workgroups_on_distinct_simds": round(breadth, 3),
        "ideal_span_if_balanced_us": round(float(busy.sum() / n), 2),
        "shader_clock_mhz_from_cycles": round(float(cyc.sum() / ((end - start).sum() + 1e-9)), 1) if sclk_mhz is None else sclk_mhz,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--gaussians", type=int, default=600_000)
    ap.add_argument("--render-res", type=int, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wg_trace.npz"))
    ap.add_argument("--hint", type=int, default=None)
    ap.add_argument("--attributes", default="trained", help="'untrained': every scale at the 0.01 m clamp (the large-splat regime)")
    ap.add_argument("--no-priority", action="store_true", help="render without GSR_FLAG_WAVE_PRIORITY (the session's default is on)")
    args = ap.parse_args()
    import torch
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import _capi, synthetic as S
    from gps_gaussian_amd import rasterizer as RZ
    from gps_gaussian_amd.session import RasterSession
    dev = torch.device("cuda:0")
    rr = args.render_res or args.res
    smp = S.make_stereo_sample(args.res, args.gaussians, seed=S.SEED, render_res=rr, attributes=args.attributes)
    g, cam = S.compact_sample(smp), smp["novel_view"]
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    P = t["means3D"].shape[0]
    sess = RasterSession(P, rr, rr, dev, training=True, wave_priority=not args.no_priority)
    a = (t["means3D"], t["colors"], t["opacities"].reshape(-1), t["scales"], t["rotations"], torch.from_numpy(cam["world_view_transform"]).to(dev),
         torch.from_numpy(cam["full_proj_transform"]).to(dev), torch.zeros(3, device=dev), math.tan(float(cam["FovX"]) * 0.5), math.tan(float(cam["FovY"]) * 0.5), 1.0)
    gout = torch.randn(3, rr, rr, device=dev)
    for _ in range(5):
        sess.forward(*a); sess.backward(gout)
    torch.cuda.synchronize()
    if args.hint is not None:
        RZ._dev_state(dev)["longest"] = args.hint
        real = RZ._learn
        RZ._learn = lambda st, R, need, P_, longest=None: real(st, R, need, P_, None)  # keep the forced hint
    NB = (((rr + 7) // 8 + 3) // 4 * 4) * ((rr + 7) // 8)
    rows = torch.zeros((2 * NB, 4), dtype=torch.int64, device=dev)
    lib = _capi.lib()
    sclk = _capi.measure_sclk_mhz(dev)
    _capi.check(lib.gsr_debug_set_wg_trace(rows.data_ptr()), "gsr_debug_set_wg_trace")
    sess.forward(*a); sess.backward(gout)
    torch.cuda.synchronize()
    _capi.check(lib.gsr_debug_set_wg_trace(None), "gsr_debug_set_wg_trace")
    h = rows.cpu().numpy().astype(np.uint64)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, fwd=h[:NB], bwd=h[NB:])
    out = {"workload": "%dx%d, P=%d, hint=%s" % (rr, rr, P, args.hint if args.hint is not None else "learnt (%s)" % RZ._dev_state(dev).get("longest")),
           "wave_priority": not args.no_priority,
           "shader_clock_mhz_under_load": round(sclk, 1),
           "fwd": analyse(h[:NB], "k_composite_fwd_tiles", None), "bwd": analyse(h[NB:], "k_composite_bwd_tiles", None)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
