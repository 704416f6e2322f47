"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (name, calls, total/avg/min/max us, %).
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc"
                     % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
    for n, k, s, a, mn, mx in rows:
        n = n if len(n) < 90 else n[:87] + "..."
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (n, k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
