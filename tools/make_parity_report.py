"""gpurun_out/parity_report.jsonl (written by tests/conftest.py::parity_report during `pytest -m gpu`) -> profiles/<tag>_parity_report.md
Usage: python tools/make_parity_report.py [tag]"""
import json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = [json.loads(l) for l in open(os.path.join(root, "gpurun_out", "parity_report.jsonl"))]
L = ["# Parity report of the GPU suite (`python -m pytest tests -m gpu`, MI355X): HIP path vs the fp32 C oracle on the same inputs.",
     "# Source: gpurun_out/parity_report.jsonl (tests/conftest.py::parity_report), formatted by tools/make_parity_report.py.",
     "# fragile = pixels within 1e-5 (relative) of an alpha / transmittance threshold in the oracle; 'touching' = Gaussians whose footprint covers one.",
     "# Gradient columns: largest |error| over Gaussians NOT touching a fragile pixel (max over the six gradient tensors), and how many of ALL /",
     "# of the untouched Gaussians are off by more than 1e-3 in any tensor.  'config2' / 'config5' / 'config2_hr' are the full-size BASELINE configs.", "",
     "| test | pixels | fragile px | RGB max err | px over 1e-4 | Gaussians | touching fragile | grad max err (untouched) | over 1e-3 (all) | over 1e-3 (untouched) | clamp-active |",
     "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    g = r.get("grads") or {}
    mx = max([v["max_err_untouched"] for v in g.values()], default=float("nan"))
    oa = max([v["gaussians_over_tol"] for v in g.values()], default=0)
    ou = max([v["gaussians_over_tol_untouched"] for v in g.values()], default=0)
    ca = r.get("clamp_active_frac_of_contributing")
    L.append("| %s | %d | %d | %.2e | %d | %d | %d | %.2e | %d | %d | %s |" % (
        r["test"], r["pixels"], r["fragile_pixels"], r["rgb_max_err"], r["pixels_over_tol"], r["gaussians"], r["gaussians_touching_fragile"], mx, oa, ou,
        ("%.2f" % ca) if ca is not None else ""))
open(os.path.join(root, "profiles", "%s_parity_report.md" % tag), "w").write("\n".join(L) + "\n")
print("\n".join(L[:12]))
