#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration, share) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db).  Prints a table and
optionally writes it as CSV:   python tools/rocpd_stats.py gpurun_out/prof/bench_results.db profiles/x.csv"""
import csv
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = db.execute(f"select s.{name_col}, d.end - d.start from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, dur in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name or "?")
        name = re.sub(r"\(.*\)$", "", name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((name, a[0], round(a[1] / 1e3, 1), round(a[1] / a[0] / 1e3, 2), round(a[2] / 1e3, 2),
                    round(a[3] / 1e3, 2), round(100.0 * a[1] / total, 2)))
    for r in out:
        print("%-90s %7s %11s %9s %9s %9s %6s" % tuple(str(x)[:90] for x in r))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)


if __name__ == "__main__":
    main()
