"""Turn the rocprofv3 output of tools/prof_r04.sh into the tracked evidence files:
   gpurun_out/prof_<tag>/{kernel_stats.md, pmc_summary.md, pmc_traffic.json}   (copy them to profiles/ to commit)
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
    # ---- pmc
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in sorted(glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(fn)):
            agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["# rocprofv3 --pmc (separate passes: SQ_*, FETCH_SIZE, WRITE_SIZE; no trace domains), mean per launch, the ONE-VIEW-IN-FLIGHT command of %s_kernel_stats_one_view.md" % tag,
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
    out = {"source": "profiles/%s_pmc_summary.md (rocprofv3 --pmc, separate passes on the one-view-in-flight command, tools/prof_%s.sh)" % (tag, tag)}
    if bench:
        import re
        m = re.search(r"(\d+)x(\d+) render.*P=(\d+) Gaussians, R=(\d+)", bench["config"]["workload"])
        if m:
            out["workload"] = {"W": int(m.group(1)), "H": int(m.group(2)), "P": int(m.group(3)), "R": int(m.group(4))}
    names = {"k_composite_bwd": "composite_bwd", "k_composite_fwd": "composite_fwd", "k_composite_bwd_tiles": "composite_bwd", "k_composite_fwd_tiles": "composite_fwd",
             "k_preprocess": "preprocess", "k_preprocess_bwd": "preprocess_bwd", "k_scatter": "scatter", "k_sort_wave": "sort"}
    for k0, v in agg.items():
        k = k0.split("<")[0]  # template instances: the one-view command only runs <true> (all six gradients / state kept); <false> never overrides it
        if k0.endswith("<false>") and names.get(k) in out:
            continue
        if k in names and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            e = {"kernel": k0, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1)}
            if "SQ_INSTS_VALU" in v:
                e["valu_wave_instructions_per_launch"] = sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"])
            out[names[k]] = e
    json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(d, "kernel_stats.md")).read()[:2500])
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
