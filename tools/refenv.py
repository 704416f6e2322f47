"""Harness around the UNMODIFIED reference scripts (tools/launch_stage2.py, tools/run_reference.py, the -m gpu reference tests).

Nothing here is product code and nothing of the reference is edited.  What it provides:
  * `reference_dir()`: where the reference lives -- --reference, $GPSGS_REFERENCE or /root/reference (a checkout: nothing of the reference travels to
    the GPU box, so what needs it runs only where a checkout exists);
  * `install_shims()`: stand-ins for yacs / cv2 / tensorboard, only when the real packages are missing;
  * `pythonpath()`: the whole integration -- gps-gaussian_amd/dropin (the MI355X rasteriser / correlation sampler under the
    reference's import names) ahead of the reference on sys.path;
  * `make_workdir()`: the reference uses cwd-relative paths (`config/stage2.yaml`, `experiments/...`, file_backup's copytree of core/ lib/
    config/ gaussian_renderer/) and its YAML ships placeholder paths, so scripts run from a scratch directory that links the reference's
    directories and holds its own `config/stage2.yaml` (editing YAML is configuration, not a code change -- SURVEY.md section 7, trap ii).
"""
import importlib
import json
import os
import sys
import types
from pathlib import Path

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DROPIN = os.path.join(ROOT, "gps-gaussian_amd", "dropin")
SHIMS = os.path.join(HERE, "shims")


def reference_dir(explicit=None):
    for p in (explicit, os.environ.get("GPSGS_REFERENCE"), "/root/reference"):
        if p and os.path.isdir(p) and (os.path.exists(os.path.join(p, "train_stage2.py")) or os.path.exists(os.path.join(p, "train_stage2.pyc"))):
            return os.path.abspath(p)
    return None


def script(ref, name):
    """Path of a top-level reference script: `name.py` in a checkout, `name.pyc` in the staged build."""
    for ext in (".py", ".pyc"):
        p = os.path.join(ref, name + ext)
        if os.path.exists(p):
            return p
    raise FileNotFoundError("%s.py[c] under %s" % (name, ref))


class _SummaryWriter:
    """Scalars as JSON lines (events.jsonl) instead of TensorBoard event files."""

    def __init__(self, log_dir=None, **_):
        self.f = None
        if log_dir:
            Path(log_dir).mkdir(parents=True, exist_ok=True)
            self.f = open(os.path.join(log_dir, "events.jsonl"), "a")

    def add_scalar(self, tag, value, step=None, **_):
        if self.f:
            self.f.write(json.dumps({"tag": tag, "value": float(value), "step": None if step is None else int(step)}) + "\n")
            self.f.flush()

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


def need_shims():
    """Names of the stand-in packages this interpreter needs (real installations always win)."""
    out = []
    for name in ("yacs", "cv2"):
        try:
            importlib.import_module(name)
        except ImportError:
            out.append(name)
    return out


def install_shims():
    """Stand-ins for packages the MI355X image lacks; never shadows a real installation."""
    if need_shims() and SHIMS not in sys.path:
        sys.path.append(SHIMS)  # behind everything else
    try:
        importlib.import_module("torch.utils.tensorboard")
    except Exception:  # noqa: BLE001  (ImportError from the missing `tensorboard` package)
        m = types.ModuleType("torch.utils.tensorboard")
        m.SummaryWriter = _SummaryWriter
        sys.modules["torch.utils.tensorboard"] = m


def pythonpath(ref):
    """sys.path entries, in order: the drop-in import names, then the reference."""
    return [DROPIN, ref]


def activate(ref):
    """Make this interpreter the integrated environment: shims (if needed), drop-in ahead of the reference on sys.path."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    install_shims()
    for p in reversed(pythonpath(ref)):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)


def make_workdir(ref, work, yaml_overrides=None, yaml_name="stage2.yaml"):
    """Scratch cwd for a reference script: links to the reference's directories and scripts, and config/<yaml_name> = the reference's
    YAML with `yaml_overrides` (a nested dict) merged in.  Returns `work`."""
    import yaml

    os.makedirs(work, exist_ok=True)
    for name in os.listdir(ref):
        if name in ("config", "experiments", "__pycache__", "MANIFEST.json") or name.startswith("."):
            continue
        dst = os.path.join(work, name)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(ref, name), dst)
    cdir = os.path.join(work, "config")
    os.makedirs(cdir, exist_ok=True)
    for name in os.listdir(os.path.join(ref, "config")):
        dst = os.path.join(cdir, name)
        if name == "__pycache__" or os.path.lexists(dst):
            continue
        if name != yaml_name:
            os.symlink(os.path.join(ref, "config", name), dst)
    cfg = yaml.safe_load(open(os.path.join(ref, "config", yaml_name)))

    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict):
                merge(a.setdefault(k, {}), v)
            else:
                a[k] = v

    merge(cfg, yaml_overrides or {})
    tmp = os.path.join(cdir, yaml_name)
    if os.path.lexists(tmp):
        os.remove(tmp)
    with open(tmp, "w") as f:
        yaml.safe_dump(cfg, f, default_flow_style=None, sort_keys=False)
    return work
