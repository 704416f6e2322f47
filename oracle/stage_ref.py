#!/usr/bin/env python
"""Compile the reference's Python tree, from the sources where they lie, into oracle/_ref/GPS-Gaussian/ (TEST INFRASTRUCTURE ONLY).

    python oracle/stage_ref.py [/root/reference]          (also: `make -C oracle ref`, and __graft_entry__.build() when the checkout exists)

Why: /root/reference does not exist on the GPU box, and the north star's contract is that the reference's own `train_stage2.py`,
`test_view_interp.py`, `lib/GaussianRender.py::pts2render`, `gaussian_renderer/__init__.py::render` and `core/corr.py` run UNMODIFIED on
the HIP drop-in.  `oracle/_ref/` is the place for reference-built artefacts: git-ignored (never part of the history), not
gpurun-ignored (it travels to the GPU box like the built .so files).  What is written there is the BUILT form of the reference -- CPython
bytecode (`.pyc`, sourceless, importable and runnable: `python test_view_interp.pyc ...`) plus the reference's YAML configuration
files (data) -- never its sources; no file of /root/reference is copied into the tracked tree.

Only `tests/` (-m gpu), `tools/run_reference.py` and `bench.py`'s full-pipeline leg put oracle/_ref on sys.path, and only as the CALLER
of the product (the reference's networks and glue), never as a substitute for it: tests/test_capi.py checks that nothing under
gps-gaussian_amd/ imports from it.

`prepare_data/` (offline data-set rendering, taichi) and `train_stage1.py` (no rasteriser in stage 1) are outside the hot path and
are not staged.

Provenance / licence note: the reference (and the Inria rasteriser licence header it carries, gaussian_renderer/__init__.py:1-10) is released
for non-commercial research and evaluation use only.  The staged bytecode and YAML files are DERIVED from it: they exist solely so that the
parity tests and the full-pipeline measurement can run the reference's own code on the GPU box, they are never committed, never imported
by the product, and must not be redistributed with it; a build from a tree without /root/reference simply has no oracle/_ref (the
reference tests skip, bench.py's full_pipeline leg says "no reference").
"""
import hashlib
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "GPS-Gaussian")

PY_DIRS = ("lib", "core", "config", "gaussian_renderer")
PY_TOP = ("train_stage2.py", "test_view_interp.py", "test_real_data.py")
DATA = (("config", ".yaml"),)


def stage(ref="/root/reference", out=OUT, quiet=False):
    ref = os.path.abspath(ref)
    if not os.path.isdir(ref):
        raise SystemExit("stage_ref: no reference checkout at %s" % ref)
    if os.path.isdir(out):
        shutil.rmtree(out)
    manifest = {"reference": ref, "python": "%d.%d" % sys.version_info[:2], "files": {}}
    todo = [(p, p) for p in PY_TOP if os.path.exists(os.path.join(ref, p))]
    for d in PY_DIRS:
        for dirpath, dirnames, filenames in os.walk(os.path.join(ref, d)):
            dirnames[:] = [x for x in dirnames if x != "__pycache__"]
            for fn in sorted(filenames):
                if fn.endswith(".py"):
                    rel = os.path.relpath(os.path.join(dirpath, fn), ref)
                    todo.append((rel, rel))
    for rel, _ in todo:
        src = os.path.join(ref, rel)
        dst = os.path.join(out, rel[:-3] + ".pyc")  # legacy location: importable without the source (SourcelessFileLoader)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: what tracebacks show -- the reference path, so a failure points at the reference's file:line
        py_compile.compile(src, cfile=dst, dfile=os.path.join("<reference>", rel), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest["files"][rel] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    for d, ext in DATA:
        for fn in sorted(os.listdir(os.path.join(ref, d))):
            if fn.endswith(ext):
                os.makedirs(os.path.join(out, d), exist_ok=True)
                shutil.copyfile(os.path.join(ref, d, fn), os.path.join(out, d, fn))
                manifest["files"][os.path.join(d, fn)] = hashlib.sha256(open(os.path.join(ref, d, fn), "rb").read()).hexdigest()
    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if not quiet:
        print("stage_ref: %d reference modules compiled into %s" % (len(todo), out))
    return out


if __name__ == "__main__":
    stage(*(sys.argv[1:2]))
