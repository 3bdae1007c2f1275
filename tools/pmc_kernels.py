#!/usr/bin/env python3
"""Per-kernel, per-grid averages of the counters in a rocprofv3 --pmc counter_collection CSV.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --output-format csv -d DIR -o x -- python3 tools/conv_bench.py
    python tools/pmc_kernels.py DIR/x_counter_collection.csv [name-substring]
"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name)
        if want not in name:
            continue
        key = (name, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
        acc[key]["~duration_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[key]["~duration_us"] += 1
    for key in sorted(acc):
        print(key)
        for c in sorted(acc[key]):
            print(f"    {c:32s} {acc[key][c] / cnt[key][c]:16.1f}   (n={cnt[key][c]})")


if __name__ == "__main__":
    main()
