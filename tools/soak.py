"""GPU-box helper: soak test -- many forward+backward steps over changing scenes, image sizes and check modes, on two streams,
to shake out rare hangs (spin-waits of the fused scan / the early capacity notification) and capacity-regrowth paths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, rasterizer as RZ

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(0)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
scenes = []
for k, (W, H, P) in enumerate([(256, 256, 30000), (640, 360, 80000), (1024, 1024, 200000), (333, 777, 50000), (2048, 1024, 150000)]):
    g = S.make_uniform_cloud(P, W, H, seed=100 + k, scale_med=0.01, z_range=(0.5, 5.0))
    t = {n: torch.from_numpy(g[n]).to(dev).requires_grad_(True) for n in ("means3D", "colors", "opacities", "scales", "rotations")}
    rs = RZ.GaussianRasterizationSettings(H, W, g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0, torch.from_numpy(g["view"]).to(dev),
                                          torch.from_numpy(g["proj"]).to(dev), 3, torch.from_numpy(g["campos"]).to(dev), False, False)
    scenes.append((t, RZ.GaussianRasterizer(rs), torch.randn(3, H, W, device=dev), torch.zeros_like(t["means3D"], requires_grad=True)))
t0 = time.time()
ref = {}
for it in range(iters):
    os.environ["GPSGS_CHECK"] = "deferred" if (it // 50) % 3 == 2 else "sync"
    RZ._early_notify = (it // 25) % 2 == 0
    k = int(rng.integers(len(scenes)))
    t, rast, gout, m2 = scenes[k]
    if it % 97 == 0 and os.environ["GPSGS_CHECK"] == "sync":
        RZ._state.clear()  # forget the learnt capacity: forces the overflow -> regrow path (deferred mode raises on overflow by design)
    if os.environ["GPSGS_CHECK"] == "deferred" and (it % 50) < len(scenes):
        os.environ["GPSGS_CHECK"] = "sync"; k_warm = it % 50  # entering a deferred phase: make sure every scene's capacity is known
        tw, rw, gw, mw = scenes[k_warm]
        with torch.no_grad():
            rw(means3D=tw["means3D"], means2D=mw, opacities=tw["opacities"], colors_precomp=tw["colors"], scales=tw["scales"], rotations=tw["rotations"])
        os.environ["GPSGS_CHECK"] = "deferred" if (it % 50) == len(scenes) - 1 else "sync"
    with torch.cuda.stream(streams[it & 1]):
        for v in t.values():
            v.grad = None
        img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        img.backward(gout)
        chk = (float(img.sum()), float(t["means3D"].grad.abs().sum()))
    if k in ref:
        assert chk == ref[k], (it, k, chk, ref[k])   # bit-reproducible across modes, streams and regrowths
    ref[k] = chk
torch.cuda.synchronize()
os.environ["GPSGS_CHECK"] = "sync"
print("SOAK_OK %d iterations in %.1f s" % (iters, time.time() - t0))
