"""Development helper (GPU box): HIP rasteriser vs the CPU oracle on synthetic scenes, with per-stage stats."""
import os, sys, time, math, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, rasterizer as RZ
from oracle.gsr_oracle import OracleRasterizer


def run_hip(g, dpix=None, debug=False):
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    for k in t: t[k].requires_grad_(True)
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                          torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                          torch.from_numpy(g["campos"]).to(dev), False, debug)
    img, radii = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"],
                                           scales=t["scales"], rotations=t["rotations"])
    grads = None
    if dpix is not None:
        img.backward(torch.from_numpy(dpix).to(dev))
        grads = {k: t[k].grad.cpu().numpy() for k in t}
        grads["means2D"] = m2.grad.cpu().numpy()
    return img.detach().cpu().numpy(), radii.cpu().numpy(), grads, img.grad_fn


def compare(name, g, with_grad=True, f64=True):
    rng = np.random.default_rng(0)
    H, W = g["H"], g["W"]
    dpix = rng.standard_normal((3, H, W)).astype(np.float32) if with_grad else None
    img, radii, grads, fn = run_hip(g, dpix)
    o = OracleRasterizer("f32")
    t0 = time.time()
    oimg, oradii = o.forward(g["means3D"], g["colors"], g["opacities"], g["scales"], g["rotations"], g["view"], g["proj"], W, H, g["tanfovx"], g["tanfovy"], g["bg"])
    t_of = time.time() - t0
    frag = o.fragility()
    err = np.abs(img - oimg).max(0)
    solid = frag > 1e-4
    res = dict(name=name, P=int(g["means3D"].shape[0]), R=o.num_rendered, radii_equal=bool((radii == oradii).all()),
               radii_mismatch=int((radii != oradii).sum()), img_maxerr=float(err.max()), img_maxerr_solid=float(err[solid].max() if solid.any() else 0),
               n_over_1e4=int((err > 1e-4).sum()), n_fragile=int((~solid).sum()), oracle_fwd_s=round(t_of, 3))
    if with_grad:
        geom = o.geom()
        # Gaussians whose footprint covers a fragile pixel may legitimately differ (branch flip at that pixel)
        fy, fx = np.nonzero(~solid)
        touched = np.zeros(res["P"], bool)
        for y, x in zip(fy, fx):
            touched |= (np.abs(geom["xy"][:, 0] - x) <= geom["radii"] + 1) & (np.abs(geom["xy"][:, 1] - y) <= geom["radii"] + 1) & (geom["radii"] > 0)
        res["n_touched"] = int(touched.sum())
        refs = {"f32": o.backward(dpix)}
        if f64:
            of = OracleRasterizer("f64")
            of.forward(g["means3D"], g["colors"], g["opacities"], g["scales"], g["rotations"], g["view"], g["proj"], W, H, g["tanfovx"], g["tanfovy"], g["bg"], decisions=geom)
            refs["f64d"] = of.backward(dpix)
        for rn, og in refs.items():
            for k in grads:
                d = np.abs(grads[k] - og[k]); s_ = np.abs(og[k]).max() + 1e-30
                rel_el = d / (np.abs(og[k]) + 1e-3 * s_)
                ok = ~touched
                res["g_%s_%s" % (rn, k)] = "max|d|/scale=%.2e relel_max(untouched)=%.2e n(rel>1e-3)=%d of which untouched %d" % (
                    d.max() / s_, rel_el[ok].max() if ok.any() else 0, int((rel_el > 1e-3).any(axis=-1).sum()), int((rel_el[ok] > 1e-3).any(axis=-1).sum()))
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "cloud", "c2"]
    if "tiny" in which:
        compare("tiny", S.make_uniform_cloud(200, 48, 40, seed=1, scale_med=0.05, z_range=(0.3, 4.0)))
    if "c1" in which:
        compare("C1 256^2 30k", S.make_scene(256, 30000))
    if "cloud" in which:
        compare("cloud 20k 333x277", S.make_uniform_cloud(20000, 333, 277, seed=3, scale_med=0.02))
    if "c2" in which:
        g = S.make_scene(1024, 600000)
        compare("C2 1024^2 600k", g, f64=False)
        # timing
        dev = torch.device("cuda:0")
        t = {k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev).requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
        m2 = torch.zeros_like(t["means3D"], requires_grad=True)
        rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                              torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                              torch.from_numpy(g["campos"]).to(dev), False, False)
        rast = RZ.GaussianRasterizer(rs)
        gout = torch.randn(3, g["H"], g["W"], device=dev)
        for mode in ("fwd", "fwd+bwd"):
            for it in range(3 + 20):
                if it == 3:
                    torch.cuda.synchronize(); t0 = time.time()
                if mode == "fwd":
                    with torch.no_grad():
                        img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
                else:
                    img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
                    img.backward(gout)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 20
            print("timing %s: %.3f ms/view  (%.1f views/s)" % (mode, dt * 1e3, 1 / dt))
        print(RZ.last_stats())
