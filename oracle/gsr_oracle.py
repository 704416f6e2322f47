"""ctypes binding of oracle/gsr_oracle.c (fp32 and fp64 builds).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see gsr_oracle.c header): restates the upstream diff_gaussian_rasterization algorithm
 at the call-site configuration of /root/reference/gaussian_renderer/__init__.py:36-62.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile the oracle shared objects with gcc (recipe: oracle/Makefile)."""
    targets = [os.path.join(_BUILD, n) for n in ("libgsr_oracle_f32.so", "libgsr_oracle_f64.so", "libaux_oracle.so")]
    srcs = [os.path.join(_HERE, n) for n in ("gsr_oracle.c", "aux_oracle.c", "Makefile")]
    stale = force or any(
        (not os.path.exists(t)) or any(os.path.getmtime(s) > os.path.getmtime(t) for s in srcs if os.path.exists(s))
        for t in targets
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B" if force else "-s", "all"], check=True)
    return targets


_libs = {}


def _lib(kind):
    if kind not in _libs:
        build()
        _libs[kind] = C.CDLL(os.path.join(_BUILD, "libgsr_oracle_%s.so" % kind))
    return _libs[kind]


class OracleRasterizer:
    """Stateful forward/backward pair, numpy in / numpy out.

    kind='f32': decision-level twin of the HIP kernels (and the CPU baseline);  kind='f64': gradient truth.
    """

    def __init__(self, kind="f32"):
        assert kind in ("f32", "f64")
        self.kind = kind
        self.np = np.float32 if kind == "f32" else np.float64
        self.ct = C.c_float if kind == "f32" else C.c_double
        self.lib = _lib(kind)
        self.pfx = "gsr_oracle_%s_" % kind
        self._f("create").restype = C.c_void_p
        self.h = C.c_void_p(self._f("create")())
        self._f("num_rendered").restype = C.c_int64
        self.args = None

    def _f(self, n):
        return getattr(self.lib, self.pfx + n)

    def __del__(self):
        try:
            self._f("destroy")(self.h)
        except Exception:
            pass

    def _a(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.np))
        if shape is not None:
            a = a.reshape(shape)
        return a

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def forward(self, means3D, colors, opacities, scales, rotations, view, proj, W, H, tanfovx, tanfovy, bg,
                scale_modifier=1.0, decisions=None, shs=None, sh_degree=0, campos=None, cov3D_precomp=None):
        """view/proj: the [4,4] tensors exactly as the reference hands them over (transposed; flat = column-major).
        decisions: optional geom() dict of ANOTHER evaluation (normally the fp32 oracle) whose discrete decisions
        (radii/visibility, tile rect, fp32 depth sort key) are adopted; see gsr_oracle.c.
        The other half of the upstream interface: shs [P,M,3] (+ sh_degree 0..3, campos [3]) INSTEAD of colors (pass colors=None),
        cov3D_precomp [P,6] INSTEAD of scales + rotations (pass both None).
        Returns (color[3,H,W], radii[P])."""
        P = int(np.asarray(means3D).reshape(-1, 3).shape[0])
        a = dict(
            means3D=self._a(means3D, (P, 3)), colors=None if colors is None else self._a(colors, (P, 3)), opacities=self._a(opacities, (P,)),
            scales=None if scales is None else self._a(scales, (P, 3)), rotations=None if rotations is None else self._a(rotations, (P, 4)),
            view=self._a(view, (16,)), proj=self._a(proj, (16,)), bg=self._a(bg, (3,)),
            shs=None if shs is None else self._a(shs, (P, -1, 3)), campos=None if campos is None else self._a(campos, (3,)),
            cov3D_precomp=None if cov3D_precomp is None else self._a(cov3D_precomp, (P, 6)),
        )
        M = 0 if a["shs"] is None else int(a["shs"].shape[1])
        self.sh = (int(sh_degree), M)
        self.args = (P, a, float(scale_modifier), int(W), int(H), float(tanfovx), float(tanfovy))
        out = np.zeros((3, H, W), self.np)
        radii = np.zeros((max(P, 1),), np.int32)
        ov = (None, None, None)
        if decisions is not None and P > 0:
            ov = (np.ascontiguousarray(decisions["radii"], np.int32), np.ascontiguousarray(decisions["rect"], np.int32),
                  np.ascontiguousarray(decisions["depth"], np.float32))
        self._ov = ov
        rc = self._f("forward_ex")(
            self.h, C.c_int(P), self._p(a["means3D"]), self._p(a["colors"]), self._p(a["opacities"]),
            self._p(a["scales"]), self._p(a["rotations"]), self.ct(scale_modifier), self._p(a["view"]),
            self._p(a["proj"]), C.c_int(W), C.c_int(H), self.ct(tanfovx), self.ct(tanfovy), self._p(a["bg"]),
            self._p(out), self._p(radii), self._p(ov[0]), self._p(ov[1]), self._p(ov[2]),
            self._p(a["shs"]), C.c_int(self.sh[0]), C.c_int(M), self._p(a["campos"]), self._p(a["cov3D_precomp"]))
        if rc != 0:
            raise RuntimeError("oracle forward failed rc=%d" % rc)
        return out, radii[:P]

    def backward(self, dL_dpix, debug=False):
        """-> dict of gradients: means3D, means2D, colors (dL/d of the blended colour, also with SH inputs), opacities, scales, rotations;
        with SH inputs also shs [P,M,3], with cov3D_precomp also cov3D_precomp [P,6] (scales / rotations then stay zero)."""
        P, a, mod, W, H, tx, ty = self.args
        g = self._a(dL_dpix, (3, H, W))
        n = max(P, 1)
        deg, M = self.sh
        o = dict(means3D=np.zeros((n, 3), self.np), means2D=np.zeros((n, 3), self.np), colors=np.zeros((n, 3), self.np),
                 opacities=np.zeros((n, 1), self.np), scales=np.zeros((n, 3), self.np), rotations=np.zeros((n, 4), self.np))
        dconic = np.zeros((n, 3), self.np) if debug else None
        dcov = np.zeros((n, 6), self.np) if (debug or a["cov3D_precomp"] is not None) else None
        dsh = np.zeros((n, M, 3), self.np) if a["shs"] is not None else None
        rc = self._f("backward_ex")(
            self.h, C.c_int(P), self._p(a["means3D"]), self._p(a["colors"]), self._p(a["opacities"]),
            self._p(a["scales"]), self._p(a["rotations"]), self.ct(mod), self._p(a["view"]), self._p(a["proj"]),
            C.c_int(W), C.c_int(H), self.ct(tx), self.ct(ty), self._p(a["bg"]), self._p(g),
            self._p(o["means3D"]), self._p(o["means2D"]), self._p(o["colors"]), self._p(o["opacities"]),
            self._p(o["scales"]), self._p(o["rotations"]), self._p(dconic), self._p(dcov),
            self._p(a["shs"]), C.c_int(deg), C.c_int(M), self._p(a["campos"]), self._p(a["cov3D_precomp"]), self._p(dsh))
        if rc != 0:
            raise RuntimeError("oracle backward failed rc=%d" % rc)
        o = {k: v[:P] for k, v in o.items()}
        if debug:
            o["conic"] = dconic[:P]
            o["cov3D"] = dcov[:P]
        if a["shs"] is not None:
            o["shs"] = dsh[:P]
        if a["cov3D_precomp"] is not None:
            o["cov3D_precomp"] = dcov[:P]
        return o

    def rgb(self):
        """[P,3] colours that were blended (precomputed, or evaluated from the SH coefficients); zeros for invisible Gaussians."""
        P = self.args[0]
        out = np.zeros((max(P, 1), 3), self.np)
        self._f("export_rgb")(self.h, self._p(out))
        return out[:P]

    @property
    def num_rendered(self):
        return int(self._f("num_rendered")(self.h))

    def geom(self):
        P, _, _, W, H, _, _ = self.args
        n = max(P, 1)
        d = dict(depth=np.zeros(n, self.np), xy=np.zeros((n, 2), self.np), conic_opacity=np.zeros((n, 4), self.np),
                 cov3D=np.zeros((n, 6), self.np), radii=np.zeros(n, np.int32), tiles=np.zeros(n, np.uint32),
                 rect=np.zeros((n, 4), np.int32))
        self._f("export_geom")(self.h, *[self._p(d[k]) for k in ("depth", "xy", "conic_opacity", "cov3D", "radii", "tiles", "rect")])
        return {k: v[:P] for k, v in d.items()}

    def binning(self):
        P, _, _, W, H, _, _ = self.args
        gx, gy = (W + 15) // 16, (H + 15) // 16
        R = self.num_rendered
        d = dict(point_list=np.zeros(max(R, 1), np.uint32), ranges=np.zeros((gx * gy, 2), np.int64),
                 final_T=np.zeros((H, W), self.np), n_contrib=np.zeros((H, W), np.uint32))
        self._f("export_binning")(self.h, self._p(d["point_list"]), self._p(d["ranges"]), self._p(d["final_T"]), self._p(d["n_contrib"]))
        d["point_list"] = d["point_list"][:R]
        return d

    def fragility(self, power_band=1e-6):
        """[H,W] margin map: how close each pixel came to one of the discontinuous branch thresholds (alpha = 1/255, T = 1e-4, and --
        unless power_band = 0 -- an exponent within power_band of upstream's `power > 0` skip)."""
        P, _, _, W, H, _, _ = self.args
        m = np.ones((H, W), self.np)
        f = self._f("fragility_ex")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        f(self.h, self._p(m), float(power_band))
        return m

    def flip_bound(self, dL_dpix, thresh=1e-5, power_band=1e-6):
        """-> (touched [P] bool, bounds dict shaped like backward()'s gradients).  touched[g]: Gaussian g takes part (alpha >= 1/255 with 1 % slack, inside
        its tile rect, in front of a clear stop) in a pixel that sits within `thresh` of a branch threshold; bounds[k][g]: upper bound on the magnitude
        of the part of g's gradient k that comes from those pixels -- what a flipped branch decision there can change (gsr_oracle.c flip_bound)."""
        P, a, mod, W, H, tx, ty = self.args
        g = self._a(dL_dpix, (3, H, W))
        margin = self.fragility(power_band)
        n = max(P, 1)
        deg, M = self.sh
        touched = np.zeros(n, np.uint8)
        use_cov, use_sh = a["cov3D_precomp"] is not None, a["shs"] is not None
        b = dict(means3D=np.zeros((n, 3), self.np), means2D=np.zeros((n, 3), self.np), colors=np.zeros((n, 3), self.np), opacities=np.zeros((n, 1), self.np),
                 scales=None if use_cov else np.zeros((n, 3), self.np), rotations=None if use_cov else np.zeros((n, 4), self.np),
                 cov3D_precomp=np.zeros((n, 6), self.np) if use_cov else None, shs=np.zeros((n, M, 3), self.np) if use_sh else None)
        f = self._f("flip_bound")
        rc = f(self.h, C.c_int(P), self._p(a["means3D"]), self._p(a["opacities"]), self._p(a["scales"]), self._p(a["rotations"]), self.ct(mod),
               self._p(a["view"]), self._p(a["proj"]), C.c_int(W), C.c_int(H), self.ct(tx), self.ct(ty), self._p(a["bg"]), self._p(g),
               self._p(a["shs"]), C.c_int(deg), C.c_int(M), self._p(a["campos"]), self._p(a["cov3D_precomp"]),
               self._p(margin), self.ct(thresh), C.c_double(power_band), touched.ctypes.data_as(C.c_void_p),
               self._p(b["means3D"]), self._p(b["means2D"]), self._p(b["colors"]), self._p(b["opacities"]), self._p(b["scales"]), self._p(b["rotations"]),
               self._p(b["cov3D_precomp"]), self._p(b["shs"]))
        if rc != 0:
            raise RuntimeError("oracle flip_bound failed rc=%d" % rc)
        return touched[:P].astype(bool), {k: v[:P] for k, v in b.items() if v is not None}


def mark_visible(means3D, view):
    """Upstream checkFrustum / markVisible: point i is visible iff its view-space depth exceeds 0.2 -- the near-plane test
    gsr_oracle.c applies (`pv[2] <= 0.2 -> culled`, xf4x3's evaluation order, fp32).  view: flat column-major [16] (the transposed tensor the
    reference passes).  -> bool [P]."""
    p = np.asarray(means3D, np.float32).reshape(-1, 3)
    m = np.asarray(view, np.float32).reshape(-1)
    z = ((m[2] * p[:, 0] + m[6] * p[:, 1]).astype(np.float32) + m[10] * p[:, 2]).astype(np.float32) + m[14]
    return z.astype(np.float32) > np.float32(0.2)

