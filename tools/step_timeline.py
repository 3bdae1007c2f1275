#!/usr/bin/env python3
"""Timeline of the last complete training step in a rocprofv3 kernel trace (CSV):

    rocprofv3 --kernel-trace --output-format csv -d DIR -o b -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --profile-steps 0
    python tools/step_timeline.py DIR/b_kernel_trace.csv [first-kernel-substring]

A step starts at each launch of the first kernel of the step (default: logmel_db_kernel).  Prints, for the last
complete step, every kernel with its start offset, duration, queue, and the idle gap since the previous kernel on
the same queue ended; then the busy time per queue and the step length."""
import csv
import re
import sys


def main():
    first = sys.argv[2] if len(sys.argv) > 2 else "logmel_db_kernel"
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], name))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[3]]
    if len(starts) < 3:
        sys.exit("fewer than three steps in the trace")
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0 = step[0][0]
    last_end = {}
    busy = {}
    print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s} {'q':>3s}  kernel")
    for s, e, q, name in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (e - s)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {q:>3s}  {name[:80]}")
    print("step length %.1f us (start of one step to start of the next)" % ((rows[b][0] - t0) / 1e3))
    for q in sorted(busy):
        print("queue %s busy %.1f us" % (q, busy[q] / 1e3))


if __name__ == "__main__":
    main()
