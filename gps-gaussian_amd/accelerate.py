"""Opt-in: let the reference's UNMODIFIED scripts reach the fused kernels of SURVEY.md section 8(f).

The two drop-in import names (`diff_gaussian_rasterization`, `corr_sampler`) replace external extensions, so the reference reaches them by
importing.  The four "(f) next" rows replace code that lives INSIDE the reference (lib/GaussianRender.py:6-40 `pts2render`, lib/loss.py:36-83
`l1_loss` / `ssim`, core/corr.py:31-61 `CorrBlockFast1D`, core/raft_stereo_human.py:69-81 `upsample_flow`, lib/utils.py:88-120
`flow2depth` / `depth2pc`), so an unmodified `train_stage2.py` / `test_view_interp.py` never calls them -- unless asked to:

    GPSGS_ACCELERATE=pack,loss,corr,upsample,unproject   (or "all"; unset / empty = off, the default)

With the variable set, importing either drop-in installs a `sys.meta_path` finder that lets the reference's modules load from where they lie,
untouched, and -- after each one has executed -- rebinds the names above to the fused implementations of this package: in the defining module
and in every loaded module that had already bound the same object with `from ... import ...` (train_stage2.py:16-17 bind `pts2render`,
`l1_loss`, `ssim` after `lib.network` -> `core.corr` has imported `corr_sampler` at :13, i.e. after this hook is in place).  No reference file
changes on disk; nothing here runs unless the variable is set; every replacement is the GPU path of this package (no CPU fallback).

`installed()` reports what was rebound, `calls` counts how often each replacement ran (the tests assert the kernels were reached).
"""
import importlib.abc
import os
import sys
import threading

FEATURES = ("pack", "loss", "corr", "upsample", "unproject")
_PREFIXES = ("lib", "core", "gaussian_renderer", "config", "train_stage2", "test_view_interp", "test_real_data")

calls = {f: 0 for f in FEATURES}
# calls that went THROUGH a replacement but were handed on to the reference's own function (a configuration the fused kernel does not implement):
# counted, so that a run on the eager path for those calls leaves a trace; the full-pipeline runner prints them
calls["loss_passthrough"] = 0
calls["unproject_passthrough"] = 0
_installed = {}          # "module.attr" -> feature
_originals = {}          # "module.attr" -> (holder, name, the reference's own object): restore(), and the two pass-through cases below
_armed = False           # set by install(); cleared once every requested name is rebound (late_apply() is then a flag test)
_lock = threading.RLock()
_finder = None


def requested(value=None):
    """The feature set GPSGS_ACCELERATE asks for (an unknown name raises: a typo must not silently run the eager path)."""
    v = os.environ.get("GPSGS_ACCELERATE", "") if value is None else value
    v = v.strip().lower()
    if v in ("", "0", "off", "none", "false"):
        return ()
    if v in ("1", "all", "on", "true"):
        return FEATURES
    names = tuple(x.strip() for x in v.replace(";", ",").split(",") if x.strip())
    bad = [x for x in names if x not in FEATURES]
    if bad:
        raise ValueError("GPSGS_ACCELERATE: unknown feature(s) %s (known: %s)" % (bad, ", ".join(FEATURES)))
    return tuple(f for f in FEATURES if f in names)


def installed():
    return dict(_installed)


# ---- the replacements (thin: argument shapes of the reference on the outside, this package's fused ops inside) -------------------------
def _pts2render(data, bg_color):
    """lib/GaussianRender.py:6 -- same contract; fused pack + one autograd node for the batch (render_api.pts2render)."""
    from . import render_api
    calls["pack"] += 1
    return render_api.pts2render(data, bg_color)


def _l1_loss(network_output, gt):
    from . import loss
    calls["loss"] += 1
    return loss.l1_and_ssim_shared(network_output, gt)[0]


def _ssim(img1, img2, window_size=11, size_average=True):
    from . import loss
    if window_size != 11 or not size_average:
        # not the configuration the fused kernel implements: the reference's own function (still on the GPU, still the reference's code)
        calls["loss_passthrough"] += 1
        return _originals["lib.loss.ssim"][2](img1, img2, window_size, size_average)
    calls["loss"] += 1
    return loss.l1_and_ssim_shared(img1, img2)[1]


def _upsample_flow(self, flow, mask):
    """core/raft_stereo_human.py:69 (a method of FlowUpdateModule)."""
    from . import corr
    calls["upsample"] += 1
    return corr.upsample_flow(flow, mask, 2 ** self.args.n_downsample)


def _flow2depth(data):
    """lib/utils.py:113 -- returns the inverse-depth map; the SAME launch also produced the world points, which ride on the returned tensor
    until lib/network.py:67 asks for them through depth2pc (the reference calls the two back to back, lib/network.py:66-67)."""
    from . import unproject
    calls["unproject"] += 1
    depth, xyz, _valid = unproject.unproject(data['flow_pred'], data['mask'], data['ref_intr'], data['intr'], data['extr'], data['Tf_x'])
    depth._gpsgs_xyz = (xyz, data['extr'], data['intr'])
    return depth


def _depth2pc(depth, extrinsic, intrinsic):
    """lib/utils.py:88 -- the points computed together with `depth` when it came from _flow2depth with these cameras; any other depth map goes
    through the reference's own function."""
    stash = getattr(depth, "_gpsgs_xyz", None)
    if stash is not None and stash[1] is extrinsic and stash[2] is intrinsic:
        return stash[0]
    calls["unproject_passthrough"] += 1
    return _originals["lib.utils.depth2pc"][2](depth, extrinsic, intrinsic)


_corr_cls = None


def _corr_block():
    """core/corr.py:31 -- this package's CorrBlockFast1D (fused volume + pyramid build, all levels sampled in one launch), counting its uses."""
    global _corr_cls
    if _corr_cls is None:
        from . import corr

        class CorrBlockFast1D(corr.CorrBlockFast1D):
            def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
                calls["corr"] += 1
                super().__init__(fmap1, fmap2, num_levels=num_levels, radius=radius)

        _corr_cls = CorrBlockFast1D
    return _corr_cls


def _table(features):
    """feature -> [(defining module, attribute or 'Class.attr', replacement)]."""
    t = {f: [] for f in features}
    if "pack" in t:
        t["pack"].append(("lib.GaussianRender", "pts2render", _pts2render))
    if "loss" in t:
        t["loss"] += [("lib.loss", "l1_loss", _l1_loss), ("lib.loss", "ssim", _ssim)]
    if "corr" in t:
        t["corr"].append(("core.corr", "CorrBlockFast1D", _corr_block()))
    if "upsample" in t:
        t["upsample"].append(("core.raft_stereo_human", "FlowUpdateModule.upsample_flow", _upsample_flow))
    if "unproject" in t:
        t["unproject"] += [("lib.utils", "flow2depth", _flow2depth), ("lib.utils", "depth2pc", _depth2pc)]
    return t


def _loaded(name):
    """A module that has finished executing (a module still inside its own import is in sys.modules too)."""
    m = sys.modules.get(name)
    if m is None:
        return None
    spec = getattr(m, "__spec__", None)
    if spec is not None and getattr(spec, "_initializing", False):
        return None
    return m


def apply(features=None):
    """Rebind whatever can be rebound NOW (idempotent; called after every reference module finishes loading).  Returns the number of new
    bindings made."""
    feats = requested() if features is None else tuple(features)
    if not feats:
        return 0
    made = 0
    with _lock:
        for feat, rows in _table(feats).items():
            for modname, attr, repl in rows:
                mod = _loaded(modname)
                if mod is None:
                    continue
                key = modname + "." + attr
                holder, name = mod, attr
                if "." in attr:
                    cls, name = attr.split(".")
                    holder = getattr(mod, cls, None)
                    if holder is None:
                        continue
                cur = holder.__dict__.get(name) if isinstance(holder, type) else getattr(holder, name, None)
                if cur is None or cur is repl:
                    continue
                if key not in _originals:
                    _originals[key] = (holder, name, cur)
                orig = _originals[key][2]
                setattr(holder, name, repl)
                if _installed.get(key) is None:
                    _installed[key] = feat
                    made += 1
                if holder is mod:
                    # names other loaded modules bound with `from <modname> import <attr>` before this ran (also `__main__`)
                    for other in list(sys.modules.values()):
                        d = getattr(other, "__dict__", None)
                        if d is not None and other is not mod and d.get(name) is orig:
                            d[name] = repl
                            _installed["%s.%s" % (getattr(other, "__name__", "?"), name)] = feat
    if made and os.environ.get("GPSGS_ACCELERATE_VERBOSE"):
        print("[gps_gaussian_amd.accelerate] rebound: " + ", ".join(sorted(_installed)), file=sys.stderr)
    return made


def restore():
    """Undo every rebinding (tests)."""
    with _lock:
        for key, (holder, name, orig) in list(_originals.items()):
            repl = holder.__dict__.get(name) if isinstance(holder, type) else getattr(holder, name, None)
            setattr(holder, name, orig)
            if not isinstance(holder, type) and repl is not None:
                for other in list(sys.modules.values()):
                    d = getattr(other, "__dict__", None)
                    if d is not None and d.get(name) is repl:
                        d[name] = orig
        _originals.clear()
        _installed.clear()
        for k in calls:
            calls[k] = 0


_late_calls = 0


def late_apply():
    """Safety net for import orders in which no reference module loads after the drop-in (nothing for the finder to hook): the shims call this
    on first use.  A flag test once everything requested is in place -- or after 16 tries: a requested feature whose module the script never imports
    (`loss` in a pure inference script that does not touch lib.loss) must not cost every later render call a table walk; the finder still catches
    the module should it load later."""
    global _armed, _late_calls
    if not _armed:
        return
    apply()
    _late_calls += 1
    feats = requested()
    want = sum(len(rows) for rows in _table(feats).values())
    if len(_originals) >= want or _late_calls >= 16:
        _armed = False


class _AfterExec(importlib.abc.Loader):
    """Delegates to the module's real loader (source or sourceless), then gives apply() a chance."""

    def __init__(self, real):
        self._real = real

    def create_module(self, spec):
        return self._real.create_module(spec)

    def exec_module(self, module):
        self._real.exec_module(module)
        spec = getattr(module, "__spec__", None)
        init = getattr(spec, "_initializing", None)
        try:
            if init:
                spec._initializing = False   # the body HAS run; the import machinery clears the flag only after we return
            apply()
        finally:
            if init:
                spec._initializing = init

    def __getattr__(self, name):   # get_code / get_source / is_package / get_filename ... of the real loader (runpy, inspect, linecache)
        return getattr(self._real, name)


class _Finder(importlib.abc.MetaPathFinder):
    """Finds the reference's modules with the ordinary finders and wraps their loader; every other import passes through untouched."""

    def find_spec(self, name, path=None, target=None):
        if name.split(".", 1)[0] not in _PREFIXES:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, "find_spec"):
                continue
            spec = f.find_spec(name, path, target)
            if spec is not None:
                if spec.loader is not None and hasattr(spec.loader, "exec_module") and not isinstance(spec.loader, _AfterExec):
                    spec.loader = _AfterExec(spec.loader)
                return spec
        return None


def install():
    """Called by the drop-in shims on import.  Off (a no-op) unless GPSGS_ACCELERATE names features."""
    global _finder, _armed
    feats = requested()
    if not feats:
        return ()
    with _lock:
        if _finder is None:
            _finder = _Finder()
            sys.meta_path.insert(0, _finder)
        _armed = True
    apply(feats)
    return feats


def uninstall():
    global _finder, _armed, _late_calls
    with _lock:
        _armed = False
        _late_calls = 0
        if _finder is not None and _finder in sys.meta_path:
            sys.meta_path.remove(_finder)
        _finder = None
    restore()
