#!/usr/bin/env python
"""Development aid: look for rare long steps in a steady loop of C-ABI sessions (MI355X).

    python tools/stall_probe.py [--res 1024] [--render-res 2048] [--gaussians 600000] [--steps 3000] [--inflight 1] [--kind infer|train]

bench.py's `configs` block twice showed one timed block of 30 forward-only steps at config 2 / 2048^2 that took ~46 ms instead of ~6 ms.
This records the host wall time of every step (no extra synchronisation) together with what the host state did during it (workspace
reallocations, pinned-ring growth, capacity changes) and prints the outliers."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--render-res", type=int, default=2048)
    ap.add_argument("--gaussians", type=int, default=600_000)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--inflight", type=int, default=1)
    ap.add_argument("--kind", default="infer")
    args = ap.parse_args()
    import torch
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import synthetic as S
    from gps_gaussian_amd import rasterizer as RZ
    from gps_gaussian_amd.session import RasterSession
    dev = torch.device("cuda:0")
    rr = args.render_res
    smp = S.make_stereo_sample(args.res, args.gaussians, seed=S.SEED + 77, render_res=rr)
    g, cam = S.compact_sample(smp), smp["novel_view"]
    lanes = []
    for i in range(args.inflight):
        t = {k: torch.from_numpy(g[k]).to(dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
        P = t["means3D"].shape[0]
        lanes.append(dict(s=RasterSession(P, rr, rr, dev, training=args.kind == "train"), st=torch.cuda.Stream(device=dev),
                          a=(t["means3D"], t["colors"], t["opacities"].reshape(-1), t["scales"], t["rotations"], torch.from_numpy(cam["world_view_transform"]).to(dev),
                             torch.from_numpy(cam["full_proj_transform"]).to(dev), torch.zeros(3, device=dev), math.tan(float(cam["FovX"]) * 0.5), math.tan(float(cam["FovY"]) * 0.5), 1.0),
                          gout=torch.randn(3, rr, rr, device=dev)))

    def group():
        for L in lanes:
            with torch.cuda.stream(L["st"]):
                L["s"].forward_begin(*L["a"])
        for L in lanes:
            with torch.cuda.stream(L["st"]):
                L["s"].forward_end()
                if args.kind == "train":
                    L["s"].backward(L["gout"])

    for _ in range(20):
        group()
    torch.cuda.synchronize()
    ring = RZ._ring(dev)
    st = RZ._dev_state(dev)
    rec = []
    t_all = time.perf_counter()
    for i in range(args.steps // args.inflight):
        a0, c0, r0 = sum(L["s"].allocations for L in lanes), [L["s"].cap for L in lanes], len(ring.chunks)
        t0 = time.perf_counter()
        group()
        dt = time.perf_counter() - t0
        rec.append((dt, sum(L["s"].allocations for L in lanes) - a0, [L["s"].cap for L in lanes] != c0, len(ring.chunks) - r0))
        if i % 200 == 199:
            torch.cuda.synchronize()  # keep the queue bounded like a timed block does
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    d = np.array([r[0] for r in rec]) * 1e3
    slow = [(int(i), round(float(d[i]), 3), int(rec[i][1]), bool(rec[i][2]), int(rec[i][3])) for i in np.argsort(d)[::-1][:8]]
    print(json.dumps({"workload": "%dx%d P=%d %s, %d in flight" % (rr, rr, lanes[0]["a"][0].shape[0], args.kind, args.inflight), "groups": len(rec),
                      "ms_per_group": {"median": round(float(np.median(d)), 4), "p99": round(float(np.quantile(d, 0.99)), 4), "max": round(float(d.max()), 3)},
                      "views_per_s_overall": round(len(rec) * args.inflight / total, 1),
                      "slowest (index, ms, ws allocations, cap changed, ring chunks added)": slow,
                      "state": {"ratio": st.get("ratio"), "floor": st.get("floor"), "longest": st.get("longest"), "big_bins": st.get("big_bins", False)}}))


if __name__ == "__main__":
    main()
