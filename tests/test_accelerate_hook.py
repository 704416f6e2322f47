"""CPU: the opt-in import hook (gps-gaussian_amd/accelerate.py) rebinds the reference's names to the fused ops when -- and only when --
GPSGS_ACCELERATE asks for it, with the reference's modules loaded UNMODIFIED from where they lie.

Each case runs in a fresh interpreter with the integration path (dropin ahead of the reference) and imports the reference's real
`train_stage2.py` / `test_view_interp.py` as modules: the import ORDER of those files (train_stage2.py:12-17: lib.human_loader, lib.network ->
core.raft_stereo_human -> core.corr -> `import corr_sampler` (the hook goes in here), ..., lib.GaussianRender, lib.loss) is what the hook has to cope
with.  The reference is the checkout in the build container; the tests skip where there is none."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import refenv  # noqa: E402

REF = refenv.reference_dir()
pytestmark = pytest.mark.skipif(REF is None, reason="no reference checkout here")

_PRELUDE = """
import os, sys
ROOT, REF = %r, %r
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refenv
refenv.activate(REF)
os.chdir(refenv.make_workdir(REF, %r))
"""


def _run(body, env_value, tmp_path):
    code = (_PRELUDE % (ROOT, REF, str(tmp_path / "work"))) + textwrap.dedent(body)
    env = dict(os.environ)
    env.pop("GPSGS_ACCELERATE", None)
    if env_value is not None:
        env["GPSGS_ACCELERATE"] = env_value
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


_CHECK_ALL = """
        import gps_gaussian_amd.accelerate as A, gps_gaussian_amd.corr as MC
        import lib.GaussianRender, lib.loss, lib.network, lib.utils, core.corr, core.raft_stereo_human as RSH
        assert lib.GaussianRender.pts2render is A._pts2render and lib.loss.l1_loss is A._l1_loss and lib.loss.ssim is A._ssim
        assert issubclass(core.corr.CorrBlockFast1D, MC.CorrBlockFast1D) and RSH.CorrBlockFast1D is core.corr.CorrBlockFast1D   # the name FlowUpdateModule.forward looks up
        assert RSH.FlowUpdateModule.__dict__["upsample_flow"] is A._upsample_flow
        assert lib.utils.flow2depth is A._flow2depth and lib.network.flow2depth is A._flow2depth and lib.network.depth2pc is A._depth2pc
        assert lib.loss.psnr.__module__ == "lib.loss" and lib.loss.sequence_loss.__module__ == "lib.loss"          # everything else is the reference's own
        assert core.corr.CorrBlock1D.__module__ == "core.corr"
"""


def test_train_stage2_binds_the_fused_ops_when_asked(tmp_path):
    out = _run("""
        import train_stage2 as T                           # the reference's script, unmodified, imported (not run)
        import gps_gaussian_amd.accelerate as A
        assert T.pts2render is A._pts2render and T.l1_loss is A._l1_loss and T.ssim is A._ssim, A.installed()
        assert T.psnr.__module__ == "lib.loss"
    """ + _CHECK_ALL + """
        feats = set(A.installed().values())
        assert feats == set(A.FEATURES), feats
        print("HOOKED", sorted(A.installed()))
    """, "all", tmp_path)
    assert "HOOKED" in out and "lib.GaussianRender.pts2render" in out


def test_view_interp_binds_the_fused_ops_when_asked(tmp_path):
    out = _run("""
        import test_view_interp as T
        import gps_gaussian_amd.accelerate as A
        assert T.pts2render is A._pts2render, A.installed()
    """ + _CHECK_ALL + """
        print("HOOKED")
    """, "pack,loss,corr,upsample,unproject", tmp_path)
    assert "HOOKED" in out


def test_default_is_off_and_a_subset_touches_only_what_it_names(tmp_path):
    out = _run("""
        import train_stage2 as T
        assert "gps_gaussian_amd.accelerate" not in sys.modules                      # not even imported
        import lib.GaussianRender, lib.loss, core.corr, core.raft_stereo_human as RSH, lib.network
        assert T.pts2render.__module__ == "lib.GaussianRender" and T.l1_loss.__module__ == "lib.loss" and T.ssim.__module__ == "lib.loss"
        assert core.corr.CorrBlockFast1D.__module__ == "core.corr" and RSH.FlowUpdateModule.upsample_flow.__module__ == "core.raft_stereo_human"
        assert lib.network.flow2depth.__module__ == "lib.utils"
        assert not any(type(f).__name__ == "_Finder" for f in sys.meta_path)
        print("UNTOUCHED")
    """, None, tmp_path)
    assert "UNTOUCHED" in out
    out = _run("""
        import train_stage2 as T
        import gps_gaussian_amd.accelerate as A
        import core.corr, core.raft_stereo_human as RSH, lib.network
        assert T.l1_loss is A._l1_loss and T.ssim is A._ssim
        assert T.pts2render.__module__ == "lib.GaussianRender" and core.corr.CorrBlockFast1D.__module__ == "core.corr"
        assert RSH.FlowUpdateModule.upsample_flow.__module__ == "core.raft_stereo_human" and lib.network.flow2depth.__module__ == "lib.utils"
        assert set(A.installed().values()) == {"loss"}
        print("SUBSET")
    """, "loss", tmp_path)
    assert "SUBSET" in out


def test_renderer_first_import_order_is_caught_by_the_sweep_and_restore_undoes_it(tmp_path):
    # a caller that imports lib.GaussianRender FIRST: the drop-in (and with it the hook) arrives while lib.GaussianRender is still executing, and the
    # caller binds the reference's pts2render; the next reference module that loads triggers the sweep over already-bound names
    out = _run("""
        from lib.GaussianRender import pts2render
        assert pts2render.__module__ == "lib.GaussianRender"
        import gps_gaussian_amd.accelerate as A
        assert A._armed
        from lib.loss import l1_loss                      # any later reference import
        import __main__
        assert __main__.pts2render is A._pts2render and l1_loss is A._l1_loss, A.installed()
        orig = A._originals["lib.GaussianRender.pts2render"][2]
        A.uninstall()
        import lib.GaussianRender, lib.loss
        assert lib.GaussianRender.pts2render is orig and __main__.pts2render is orig and lib.loss.l1_loss.__module__ == "lib.loss"
        assert not A.installed() and not any(type(f).__name__ == "_Finder" for f in sys.meta_path)
        print("SWEPT")
    """, "all", tmp_path)
    assert "SWEPT" in out


def test_a_typo_in_the_feature_list_raises(tmp_path):
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import accelerate as A
    assert A.requested("") == () and A.requested("all") == A.FEATURES and A.requested("loss, pack") == ("pack", "loss")
    with pytest.raises(ValueError):
        A.requested("pack,los")


def test_calls_handed_on_to_the_reference_are_counted(tmp_path):
    """A configuration the fused kernels do not implement goes to the reference's own function -- and leaves a trace in `calls`."""
    out = _run("""
        import train_stage2  # noqa: F401
        import torch
        import gps_gaussian_amd.accelerate as A
        import lib.loss, lib.utils
        a, b = torch.rand(1, 3, 24, 24), torch.rand(1, 3, 24, 24)
        assert A.calls["loss_passthrough"] == 0 and A.calls["unproject_passthrough"] == 0
        v = lib.loss.ssim(a, b, window_size=7)                     # not the fused configuration: the reference's eager ssim, on the CPU tensors
        assert 0.0 < float(v) < 1.0 and A.calls["loss_passthrough"] == 1 and A.calls["loss"] == 0
        m = lib.loss.ssim(a, b, size_average=False)
        assert m.numel() == 1 and A.calls["loss_passthrough"] == 2
        depth = torch.rand(1, 1, 8, 8) + 0.5
        extr = torch.eye(4)[:3].unsqueeze(0); intr = torch.eye(3).unsqueeze(0)
        pts = lib.utils.depth2pc(depth, extr, intr)                # a depth map that did not come from the fused flow2depth
        assert pts.shape == (1, 64, 3) and A.calls["unproject_passthrough"] == 1 and A.calls["unproject"] == 0
        print("PASSTHROUGH_OK")
    """, "all", tmp_path)
    assert "PASSTHROUGH_OK" in out
