"""Summarise rocprofv3 --pmc CSV output per kernel (mean per launch).  Usage: python tools/pmc_summary.py <dir-with-pmc_*> [out.md]"""
import collections
import csv
import glob
import os
import sys


def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def main(d, out=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in sorted(glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(fn)):
            agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["| kernel | launches | " + " | ".join(counters) + " |", "|---|---|" + "---|" * len(counters)]
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        lines.append("| %s | %d | " % (k, n) + " | ".join(("%.4g" % (sum(v[c]) / len(v[c]))) if c in v else "-" for c in counters) + " |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
