"""GPU box: gradient error statistics of both compositing kernel families against the fp32 and the fp64 oracle (same decisions)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import hip_render, oracle_render, touched_by_fragile
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S

TOL = 1e-3
def norm_err(a, ref):
    s = np.abs(ref).max() + 1e-30
    return np.abs(a - ref) / (np.abs(ref) + TOL * s)

scenes = {"c1_256_30k": lambda: S.make_scene(256, 30000), "cloud_333x277_20k": lambda: S.make_uniform_cloud(20000, 333, 277, seed=3, scale_med=0.02)}
for name, mk in scenes.items():
    g = mk(); H, W = g["H"], g["W"]
    dpix = np.random.default_rng(11).standard_normal((3, H, W)).astype(np.float32)
    o, oimg, oradii = oracle_render(g, "f32")
    og = o.backward(dpix)
    geom = o.geom()
    o64, _, _ = oracle_render(g, "f64", decisions=geom)
    og64 = o64.backward(dpix)
    for band in (1e-4,):
        solid, touched = touched_by_fragile(o, band)
        res = {}
        for fam in ("valu", "mfma"):
            os.environ["GPSGS_COMPOSITE"] = fam
            img, radii, grads, _ = hip_render(g, dpix)
            r = {"img_err_solid": float(np.abs(img - oimg).max(0)[solid].max())}
            for k in ("means3D", "colors", "opacities", "scales", "rotations"):
                e32 = norm_err(grads[k], og[k]); e64 = norm_err(grads[k], og64[k]); eo = norm_err(og[k], og64[k])
                r[k] = {"max32_untouched": float(e32[~touched].max()), "n_over_32": int((e32[~touched] > TOL).sum()), "q999_32": float(np.quantile(e32, .999)),
                        "max64_untouched": float(e64[~touched].max()), "q999_64": float(np.quantile(e64, .999)), "oracle32_vs_64_max_untouched": float(eo[~touched].max()),
                        "oracle32_vs_64_q999": float(np.quantile(eo, .999))}
                if k == "scales":
                    idx = np.argsort(e32.max(-1) * (~touched))[::-1][:4]
                    r[k]["worst"] = [dict(i=int(i), e32=[float(x) for x in e32[i]], e64=[float(x) for x in e64[i]], eo=[float(x) for x in eo[i]], radius=int(oradii[i]), op=float(g["opacities"][i, 0]),
                                          scale=[float(x) for x in g["scales"][i]], g=[float(x) for x in grads[k][i]], g32=[float(x) for x in og[k][i]], g64=[float(x) for x in og64[k][i]]) for i in idx]
            res[fam] = r
        print(json.dumps({"scene": name, "band": band, "touched_frac": float(touched.mean()), "res": res}))
