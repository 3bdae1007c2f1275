#!/usr/bin/env python3
"""Merge two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately because the TCC block cannot hold
both) into per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -o b -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -o b -- python3 bench.py ...
    python tools/pmc_traffic.py DIR_F/b_counter_collection.csv DIR_W/b_counter_collection.csv out.json

Counters are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests of wide coalesced reads at 64 bytes, so the read
side is also reported doubled ("corrected"); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import json
import re
import sys


def norm(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace(" ", "")


def load(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = norm(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    f_tot, f_cnt = load(sys.argv[1], "FETCH_SIZE")
    w_tot, w_cnt = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f_tot) | set(w_tot)):
        n = max(f_cnt[k], w_cnt[k], 1)
        fk = f_tot[k] / max(f_cnt[k], 1)
        wk = w_tot[k] / max(w_cnt[k], 1)
        out[k] = {"launches": n, "fetch_kib_per_launch": round(fk, 1), "write_kib_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch_raw": int((fk + wk) * 1024),
                  "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches"])[:25]:
        print(f"{k[:70]:70s} n={v['launches']:4d} fetch {v['fetch_kib_per_launch']:10.1f} KiB write "
              f"{v['write_kib_per_launch']:10.1f} KiB")


if __name__ == "__main__":
    main()
