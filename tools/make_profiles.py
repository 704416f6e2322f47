"""Turn the rocprofv3 output of tools/prof_r06.sh into the tracked evidence files:
   gpurun_out/prof_<tag>/{kernel_stats.md, kernel_stats_one_view.md, pmc_summary.md, pmc_traffic.json, regime_kernel_stats.md, regime_pmc_summary.md,
   pmc_traffic_regime.json}   (copy them to profiles/ to commit)
Usage: python tools/make_profiles.py <prof dir> <tag>"""
import collections, csv, glob, json, os, sys


def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main(d, tag):
    # ---- kernel stats: the default command (views overlapping) and the one-view-in-flight command (exclusive durations)
    def stats_table(sub, header, out_name):
        rows = []
        for fn in glob.glob(os.path.join(d, sub, "*kernel_stats.csv")):
            rows = list(csv.DictReader(open(fn)))
        if not rows:
            return
        lines = header + ["", "| kernel | calls | total_us | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
        for r in rows[:28]:
            lines.append("| %s | %s | %.1f | %.2f | %.2f | %.2f | %s |" % (kname(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        open(os.path.join(d, out_name), "w").write("\n".join(lines) + "\n")

    stats_table("trace", ["# rocprofv3 --kernel-trace --stats, `python bench.py --steps 12 --warmup 3 --repeats 5 --no-cpu-baseline --no-configs --no-full-pipeline` (the driver's command, shorter), MI355X",
                          "# bench.py renders 6 views concurrently in its `session.views_in_flight` region (one HIP stream each): the launches of different views overlap and",
                          "# time-share the chip, so a kernel's average duration here mixes its one-view-at-a-time launches (calibration, one-view timed region,",
                          "# secondary legs) with the longer overlapped ones.  The exclusive durations -- what `roofline.avg_launch_us` is -- are in",
                          "# %s_kernel_stats_one_view.md; `roofline.views_in_flight_region.avg_launch_us` is the overlapped duration." % tag], "kernel_stats.md")
    stats_table("trace1", ["# rocprofv3 --kernel-trace --stats, `python bench.py --steps 12 --warmup 3 --repeats 5 --no-cpu-baseline --no-configs --no-full-pipeline --inflight 1 --headline-only`, MI355X: ONE view in flight,",
                           "# every launch has the chip to itself.  These are the exclusive kernel durations: `roofline.avg_launch_us` of bench.py (hipEvents on",
                           "# the launch stream over its one-view timed region) and the `stages` table agree with the avg_us column below."], "kernel_stats_one_view.md")
    stats_table("regime_trace", ["# rocprofv3 --kernel-trace --stats, `python tools/stage_times.py --families tiles --steps 10 --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --seed-offset 77`, MI355X:",
                                 "# the untrained-heads regime (scales at their 0.01 m clamp: ~25 px splats, R ~ 3e7), one view at a time (bench.py's config3_regime leg renders the same scene)"], "regime_kernel_stats.md")
    names = {"k_composite_bwd": "composite_bwd", "k_composite_fwd": "composite_fwd", "k_composite_bwd_tiles": "composite_bwd", "k_composite_fwd_tiles": "composite_fwd",
             "k_preprocess": "preprocess", "k_preprocess_bwd": "preprocess_bwd", "k_scatter": "scatter", "k_sort_wave": "sort", "k_scan_b": "scan",
             "k_sort_multi": "sort_multi", "k_sort_large": "sort_large"}

    def traffic_json(agg_, workload, source, path):
        out_ = {"source": source, "workload": workload}
        for k0, v in agg_.items():
            k = k0.split("<")[0]  # template instances: <true> (all six gradients / state kept) is what the one-view command runs; <false> never overrides it
            if k not in names or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                continue
            f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            e = {"kernel": k0, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1)}
            if "SQ_INSTS_VALU" in v:
                e["valu_wave_instructions_per_launch"] = sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"])
            key = names[k]
            if key in out_ and k0.endswith("<false>"):
                continue
            if key.startswith("sort") and key != "sort":  # the multi-wave / large-list sort launches belong to the sort stage
                if "sort" in out_:
                    out_["sort"]["hbm_bytes_per_launch"] += e["hbm_bytes_per_launch"]
                    if "valu_wave_instructions_per_launch" in e:
                        out_["sort"]["valu_wave_instructions_per_launch"] = out_["sort"].get("valu_wave_instructions_per_launch", 0.0) + e["valu_wave_instructions_per_launch"]
                    out_["sort"]["kernel"] += " + " + k0
                    continue
                key = "sort"
            out_[key] = e
        json.dump(out_, open(path, "w"), indent=1)
        return out_

    # ---- the regime's counter passes
    ragg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in sorted(glob.glob(os.path.join(d, "regime_pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(fn)):
            ragg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if ragg:
        rc = sorted({c for v in ragg.values() for c in v})
        rl = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB, mean per launch) of the untrained-heads regime command of %s_regime_kernel_stats.md; traffic = 2 x FETCH + WRITE (gfx950)" % tag,
              "", "| kernel | launches | " + " | ".join(rc) + " |", "|---|---|" + "---|" * len(rc)]
        for k, v in sorted(ragg.items()):
            rl.append("| %s | %d | " % (k, max(len(x) for x in v.values())) + " | ".join(("%.4g" % (sum(v[c]) / len(v[c]))) if c in v else "-" for c in rc) + " |")
        open(os.path.join(d, "regime_pmc_summary.md"), "w").write("\n".join(rl) + "\n")
        wl = None
        for l in open(os.path.join(d, "regime_trace.log"), errors="ignore"):
            if l.startswith("{") and '"stages_us"' in l:
                j = json.loads(l)
                wl = {"W": j.get("W"), "H": j.get("H"), "P": j.get("P"), "R": j.get("R")}
        if wl:
            traffic_json(ragg, wl, "profiles/%s_regime_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/prof_%s.sh)" % (tag, tag),
                         os.path.join(d, "pmc_traffic_regime.json"))

    # ---- pmc
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in sorted(glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(fn)):
            agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["# rocprofv3 --pmc (separate passes: SQ_* x 2, FETCH_SIZE, WRITE_SIZE, TCP/TCC requests; no trace domains), mean per launch, the ONE-VIEW-IN-FLIGHT command of %s_kernel_stats_one_view.md" % tag,
             "# FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE counts 16 B/lane reads at half size (MI355X_MICROARCH.md) -> traffic = 2 x FETCH + WRITE",
             "", "| kernel | launches | " + " | ".join(counters) + " |", "|---|---|" + "---|" * len(counters)]
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        lines.append("| %s | %d | " % (k, n) + " | ".join(("%.4g" % (sum(v[c]) / len(v[c]))) if c in v else "-" for c in counters) + " |")
    open(os.path.join(d, "pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    # ---- traffic file with the workload
    bench = None
    for l in open(os.path.join(d, "trace1.log" if os.path.exists(os.path.join(d, "trace1.log")) else "trace.log"), errors="ignore"):
        if l.startswith("{") and '"metric"' in l:
            bench = json.loads(l)
    wl = None
    if bench:
        import re
        m = re.search(r"(\d+)x(\d+) render.*P=(\d+) Gaussians, R=(\d+)", bench["config"]["workload"])
        if m:
            wl = {"W": int(m.group(1)), "H": int(m.group(2)), "P": int(m.group(3)), "R": int(m.group(4))}
    out = traffic_json(agg, wl, "profiles/%s_pmc_summary.md (rocprofv3 --pmc, separate passes on the one-view-in-flight command, tools/prof_%s.sh)" % (tag, tag),
                       os.path.join(d, "pmc_traffic.json"))
    print(open(os.path.join(d, "kernel_stats.md")).read()[:2500])
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
