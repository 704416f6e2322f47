#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats of BASELINE config 4 (tools/run_reference.py ddp, the reference's unmodified Trainer) -> profiles/<tag>_config4_kernel_stats.md:
which kernels an iteration spends its GPU time in, by family -- this repository's rasteriser kernels, its fused (f) kernels (reached through the
GPSGS_ACCELERATE import hook), MIOpen / rocBLAS / ATen kernels of the PyTorch networks.

    python tools/make_config4_profile.py gpurun_out/prof_r04_config4 r04 [iterations]
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RASTER = ("k_preprocess", "k_scan_", "k_scatter", "k_sort_", "k_composite_", "k_validate_lists")
FUSED = {"k_loss_": "f2 fused L1 + SSIM", "k_pack_": "f1 fused mask-compaction + pack", "k_cv_": "f4 correlation volume + pyramid", "k_lookup_": "f4 all-level lookup",
         "k_up_": "f4 convex upsample", "k_unproject_": "f3 flow2depth + depth2pc", "k_cs_": "a14 correlation sampler (drop-in)"}


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def family(n):
    s = short(n)
    if s.startswith(RASTER):
        return "rasteriser (this repository, rows a5-a13)"
    for k, v in FUSED.items():
        if s.startswith(k):
            return "fused (f) kernels (this repository, via GPSGS_ACCELERATE)"
    if "miopen" in n.lower() or s.startswith(("igemm_", "naive_conv", "batched_transpose", "SubTensorOp", "gfx9_", "Cijk_", "Op2dTensor", "Op1dTensor", "Op3dTensor", "Op4dTensor", "Op5dTensor")):
        return "networks: MIOpen / rocBLAS (convolutions, GEMMs, layout transposes)"
    if "rocprim" in n:
        return "networks / reference glue: rocPRIM (torch boolean indexing, sorts)"
    if s.startswith("__amd_rocclr"):
        return "runtime copies / fills"
    return "networks / loss / optimizer: ATen elementwise, normalisation, reductions"


def main(d, tag, iters, which="config4"):
    rows = []
    for fn in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(fn)))
    if not rows:
        raise SystemExit("no *kernel_stats.csv under " + d)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        fam.setdefault(family(r["Name"]), []).append(r)
    unit = "iteration" if which == "config4" else "novel view"
    if which == "config3":
        L = ["# rocprofv3 --kernel-trace --stats of BASELINE config 3 as the reference runs it: `python tools/run_reference.py interp --res 1024 --samples 2 --views 5` on an MI355X -- the reference's",
             "# unmodified test_view_interp.py (bytecode build in oracle/_ref) executed twice as __main__ plus one instrumented pass: %d novel views of 2048x2048 from 1024^2 stereo pairs, real" % iters,
             "# networks with random weights (every regressed scale at the 0.01 m clamp: ~3e7 instances per view), HIP drop-in rasteriser + correlation sampler, MIOPEN_FIND_MODE=FAST, warm run.",
             "# GPU kernel time %.1f ms per view.  Round 3's profile of the same command: `profiles/r03_config3_kernel_stats.md` (k_sort_large 1,900.8 us, k_scatter 215.5, k_preprocess 160.0 per view)." % (tot / 1e6 / iters), "",
             "## Where a view's kernel time goes", "", "| family | kernels | ms per view | % of kernel time |", "|---|---|---|---|"]
    else:
      L = ["# rocprofv3 --kernel-trace --stats of BASELINE config 4 with GPSGS_ACCELERATE=all: `python tools/run_reference.py ddp --res 1024 --steps %d --batch 4 --train-samples 4 --accelerate all`" % iters,
         "# on an MI355X -- the reference's unmodified `train_stage2.Trainer` (bytecode build in oracle/_ref) through tools/launch_stage2.py at world size 1: batch 4 stereo pairs, 1024^2 -> 2048^2,",
         "# real networks with random weights (AMP), MIOPEN_FIND_MODE=FAST.  %d iterations; GPU kernel time %.1f ms per iteration (kernels of the 4 samples' rasteriser launches overlap on 4 HIP" % (iters, tot / 1e6 / iters),
         "# streams, so their avg_us are NOT exclusive durations: bench.py's `configs.config3_regime...stages_one_view_in_flight` has those).",
         "# The kernel NAMES are the evidence that the unmodified script reached the fused kernels: k_pack_*, k_loss_*, k_cv_*, k_lookup_*, k_up_*, k_unproject_* next to the rasteriser's k_*.", "",
         "## Where an iteration's kernel time goes", "", "| family | kernels | ms per iteration | % of kernel time |", "|---|---|---|---|"]
    for f, rs in sorted(fam.items(), key=lambda kv: -sum(float(r["TotalDurationNs"]) for r in kv[1])):
        t = sum(float(r["TotalDurationNs"]) for r in rs)
        L.append("| %s | %d | %.2f | %.1f |" % (f, len(rs), t / 1e6 / iters, 100 * t / tot))
    L += ["", "## This repository's kernels in that run", "", "| kernel | row | calls per %s | avg_us | ms per %s | %% |" % (unit, unit), "|---|---|---|---|---|---|"]
    for r in rows:
        s = short(r["Name"])
        if not s.startswith("k_"):
            continue
        row = "raster" if s.startswith(RASTER) else next((v for k, v in FUSED.items() if s.startswith(k)), "")
        L.append("| %s | %s | %.1f | %.1f | %.3f | %.2f |" % (s, row, int(r["Calls"]) / iters, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / iters, 100 * float(r["TotalDurationNs"]) / tot))
    L += ["", "## Top 25 kernels overall", "", "| kernel | calls | avg_us | % |", "|---|---|---|---|"]
    for r in rows[:25]:
        L.append("| %s | %s | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    out = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.md" % (tag, which))
    open(out, "w").write("\n".join(L) + "\n")
    print("\n".join(L[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8, sys.argv[4] if len(sys.argv) > 4 else "config4")
