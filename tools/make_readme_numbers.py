#!/usr/bin/env python
"""One source for the numbers README.md quotes: a bench.py JSON line (the driver's BENCH_rNN.json `parsed` record, or a file holding the line) +
profiles/<tag>_kernel_stats_one_view.md -> profiles/<tag>_numbers.md, and (with --readme) the block between the NUMBERS markers of README.md.

    python tools/make_readme_numbers.py gpurun_out/r06_final/bench.json r06 [--readme]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    txt = open(path).read()
    try:
        d = json.loads(txt)
        return d.get("parsed", d) if isinstance(d, dict) else d
    except Exception:  # noqa: BLE001
        return json.loads([l for l in txt.splitlines() if l.startswith("{") and '"metric"' in l][-1])


def main():
    src, tag = sys.argv[1], sys.argv[2]
    d = load(src)
    rf, st, s2, sess = d["roofline"], d["stages"], d.get("stage2_gradient_set") or {}, d["session"]
    L = ["Numbers of `python bench.py` on ONE MI355X (%s; source: `%s`; BASELINE config 2 = 1024x1024, 600,000 Gaussians; synthetic data; box-to-box spread of one commit is ~2 %%):" % (
        tag, os.path.relpath(src, ROOT) if os.path.isabs(src) else src), "",
         "| | rate | ms per view / iteration |", "|---|---|---|",
         "| **config 2, raster forward + backward through the reference's plugin API** (`GaussianRasterizer` + `backward()`, one view at a time, exact capacity check): `value` | **%.0f views/s** | %.4f (quartiles %.4f - %.4f) |" % (
             d["value"], d["ms_per_step"], d["ms_per_step_iqr"][0], d["ms_per_step_iqr"][1]),
         "| the same step through a C-ABI session (caller-owned buffers), one view at a time | %.0f views/s | %.4f |" % (sess["single_view_in_flight_views_per_s"], 1e3 / sess["single_view_in_flight_views_per_s"]),
         "| C-ABI sessions, %d independent views in flight on %d HIP streams | %.0f views/s | %.4f |" % (sess["views_in_flight"], sess["views_in_flight"], sess["views_in_flight_views_per_s"], sess["views_in_flight_ms_per_step"])]
    L[4] = L[4].replace("exact capacity check): `value`", "exact capacity check, compiled host path, process NOT pinned): `value`")
    if d.get("plugin_api_pinned_views_per_s"):
        aff = d.get("cpu_affinity")
        L.insert(len(L) - 2, "| the same step with the process pinned to %s CPUs of one L3 domain next to the GPU (`dist.pin_near_gpu`), this box, this run | %.0f views/s | %.4f |" % (
            aff.get("rank0_cpus") if isinstance(aff, dict) else "?", d["plugin_api_pinned_views_per_s"], 1e3 / d["plugin_api_pinned_views_per_s"]))
    if d.get("plugin_api_python_host_path_views_per_s"):
        L.insert(len(L) - 2, "| the same step through the Python host path (`GPSGS_FAST=0`: what rounds 1-5 measured), same pinning state as the line above | %.0f views/s | %.4f |" % (
            d["plugin_api_python_host_path_views_per_s"], 1e3 / d["plugin_api_python_host_path_views_per_s"]))
    cold = d.get("single_view_cold") or {}
    if cold.get("ms_cold"):
        L.append("| C-ABI session step with a 1 GiB device fill between steps (COLD: workspace and inputs evicted from the Infinity Cache) / without the fill, same event bracket | %.0f / %.0f views/s | %.4f / %.4f |" % (
            cold["single_view_cold_views_per_s"], cold["single_view_warm_same_method_views_per_s"], cold["ms_cold"], cold["ms_warm"]))
    if d.get("forward_only_views_per_s"):
        L.append("| forward only (plugin API, no_grad); `GPSGS_CHECK=deferred` forward + backward | %.0f; %.0f views/s | %.4f; %.4f |" % (
            d["forward_only_views_per_s"], d["deferred_check_views_per_s"]["fwd_bwd"], 1e3 / d["forward_only_views_per_s"], 1e3 / d["deferred_check_views_per_s"]["fwd_bwd"]))
    if s2:
        L.append("| the gradient set stage 2 differentiates (no dL/dcolour), sessions: one view at a time / %d in flight | %.0f / %.0f views/s | |" % (
            s2["views_in_flight"], s2["single_view_in_flight_views_per_s"], s2["views_per_s"]))
    for name, c in (d.get("configs") or {}).items():
        fb = c.get("fwd_bwd")
        if not fb:
            continue
        ks = list(fb)
        L.append("| %s: forward + backward one view at a time / %s; forward only | %.0f / %.0f; %.0f views/s | %.3f / %.3f |" % (
            name.split(" (")[0], ks[1].replace("_", " "), fb[ks[0]]["views_per_s"], fb[ks[1]]["views_per_s"], c["fwd_only"][ks[0]]["views_per_s"], fb[ks[0]]["ms_per_view"], fb[ks[1]]["ms_per_view"]))
    sp = d.get("stage2_path") or {}
    if sp.get("iters_per_s"):
        L.append("| stage-2 iteration WITHOUT the networks: `pts2render` (batch 4) + L1/SSIM + backward to the per-pixel maps (`stage2_path`) | %.0f iterations/s | %.3f |" % (sp["iters_per_s"], sp["ms_per_iter"]))
    fp = d.get("full_pipeline") or {}
    a, b = fp.get("config4_stage2_as_the_reference_runs_it") or {}, fp.get("config4_stage2_accelerated") or {}
    if a.get("stage2_iters_per_s") and b.get("stage2_iters_per_s"):
        L.append("| **BASELINE config 4**: the reference's unmodified `train_stage2.Trainer`, batch 4, 1024² -> 2048², real networks (random weights), MIOpen FAST find mode -- as the reference runs it / with `GPSGS_ACCELERATE=all` | %.2f / **%.2f iterations/s** | %.0f / %.0f |" % (
            a["stage2_iters_per_s"], b["stage2_iters_per_s"], a["iter_ms"], b["iter_ms"]))
    a, b = fp.get("config3_view_interp_as_the_reference_runs_it") or {}, fp.get("config3_view_interp_accelerated") or {}
    if a.get("views_per_s_within_sample") and b.get("views_per_s_within_sample"):
        L.append("| **BASELINE config 3**: the reference's unmodified `test_view_interp.py`, real networks -- as the reference runs it / hooked: views/s inside a sample (whole script) | %.1f (%.1f) / %.1f (%.1f) views/s | |" % (
            a["views_per_s_within_sample"], a["views_per_s_whole_script"], b["views_per_s_within_sample"], b["views_per_s_whole_script"]))
    cb = d.get("cpu_baseline") or {}
    if cb:
        L.append("| CPU oracle (fp32, OpenMP, %d host threads of the GPU box) forward + backward (`cpu_baseline`) | %.2f views/s | %.0f |" % (cb["cores"], cb["value"], 1e3 / cb["value"]))
    cs = d.get("cpu_taichi_splat_port") or {}
    if cs.get("value"):
        L.append("| CPU port of the reference's Taichi point splat (1 core; forward z-buffer splat only) | %.1f views/s | %.1f |" % (cs["value"], 1e3 / cs["value"]))
    L += ["", "Per kernel, one view at a time (`stages`: hipEvent brackets, each ~1-3 µs above the rocprofv3 duration of `profiles/%s_kernel_stats_one_view.md`); bytes per launch; "
          "`hbm frac` from the counter bytes where known; VALU issue = wave64 VALU instructions x 4 cycles / (1,024 SIMDs x measured clock x t):" % tag, "",
          "| kernel | µs | algorithmic MB | counter MB (2 x FETCH + WRITE) | ratio | hbm frac | VALU issue | bound |", "|---|---|---|---|---|---|---|---|"]
    for k, v in st.items():
        L.append("| %s | %.1f | %.1f | %s | %s | %s | %s | %s |" % (k, v["avg_us"], v["algorithmic_bytes"] / 1e6, ("%.1f" % (v["counter_bytes"] / 1e6)) if "counter_bytes" in v else "",
                                                                   v.get("traffic_ratio", ""), v.get("hbm_frac", ""), v.get("valu_issue_frac", ""), v.get("bound", "")))
    nb = d.get("neighbours") or {}
    if nb and "error" not in nb:
        L += ["", "Neighbouring kernels (SURVEY section 8f rows f1-f4) at BASELINE config 4's sizes: GPU time (calls queued behind a spin kernel, one event bracket), algorithmic bytes, HBM fraction:", "",
              "| op | kernels | forward µs | MB | hbm frac | backward µs | MB | hbm frac |", "|---|---|---|---|---|---|---|---|"]
        for k, v in nb.items():
            f, b = v.get("forward") or {}, v.get("backward") or {}
            L.append("| %s | %s | %s | %s | %s | %s | %s | %s |" % (k, v.get("kernels", ""), f.get("us", ""), ("%.0f" % (f["algorithmic_bytes"] / 1e6)) if f else "", f.get("hbm_frac", ""),
                                                                   b.get("us", ""), ("%.0f" % (b["algorithmic_bytes"] / 1e6)) if b else "", b.get("hbm_frac", "")))
    rg = None
    for name, c in (d.get("configs") or {}).items():
        if "regime" in name and c.get("stages_one_view_in_flight"):
            rg = c
    if rg:
        L += ["", "Untrained-heads regime (P = %d at %s, R = %d, scanned lists), per kernel, one view at a time:" % (rg["P"], rg["render"], rg["R"]), "",
              "| kernel | µs | algorithmic MB | counter MB | hbm frac | VALU issue | bound |", "|---|---|---|---|---|---|---|"]
        for k, v in rg["stages_one_view_in_flight"].items():
            L.append("| %s | %.1f | %.1f | %s | %s | %s | %s |" % (k, v["avg_us"], v["algorithmic_bytes"] / 1e6, ("%.1f" % (v["counter_bytes"] / 1e6)) if "counter_bytes" in v else "",
                                                                  v.get("hbm_frac", ""), v.get("valu_issue_frac", ""), v.get("bound", "")))
        s2t = rg.get("stages_stage2_gradient_set") or {}
        if s2t:
            L.append("")
            L.append("The two backward kernels with the gradient set stage 2 differentiates (no dL/dcolour: one-sector records that carry dL/dopacity): " +
                     ", ".join("%s %.1f µs" % (k, v["avg_us"]) for k, v in s2t.items()) + "; forward + backward %s views/s one view at a time." % (
                         (rg.get("fwd_bwd_stage2_gradient_set") or {}).get("one_view_in_flight", {}).get("views_per_s", "?")))
    L += ["", "Roofline of the dominant kernel (`%s`, %.1f µs exclusive, median block of %s launches): SURVEY §8(d) bytes %.1f MB -> %.0f GB/s = **%.3f of the 8 TB/s HBM peak**; with the implementation's own 8x8-bin instance count %.3f; "
          "by the counters %s GB/s (traffic ratio %s); VALU issue fraction %s at a measured %.0f MHz." % (
              rf["kernel"], rf["avg_launch_us"], rf.get("launches_averaged"), rf["algorithmic_bytes_per_launch"] / 1e6, rf["achieved"], rf["frac"], rf["frac_with_bin_8x8_instances"],
              rf.get("counter_gbs"), rf.get("traffic_ratio"), rf.get("valu_issue_frac"), rf.get("shader_clock_mhz") or 0)]
    txt = "\n".join(L) + "\n"
    open(os.path.join(ROOT, "profiles", "%s_numbers.md" % tag), "w").write(txt)
    if "--readme" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        s = open(p).read()
        s = re.sub(r"(<!-- NUMBERS:BEGIN -->\n).*?(<!-- NUMBERS:END -->)", lambda m: m.group(1) + txt + m.group(2), s, flags=re.S)
        open(p, "w").write(s)
    print(txt)


if __name__ == "__main__":
    main()
