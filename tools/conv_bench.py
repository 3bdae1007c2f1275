#!/usr/bin/env python3
"""Per-shape timing of the model's dense contractions (forward, data gradient, weight gradient) at batch B.
Developer aid for kernel tuning:  python tools/conv_bench.py [B]   (env MMSA_CONV_BKC=1|2|4 to force the stage depth)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sentiment-analysis_amd"))
import torch  # noqa: E402

from mmsa_amd import ops  # noqa: E402

DEV = "cuda"
PEAK = 157.3


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    shapes = [("rb1.c1", 32, 65, 32, 64, 9), ("rb1.c2", 32, 65, 64, 64, 9), ("rb1.sc", 32, 65, 32, 64, 1),
              ("rb2.c1", 16, 32, 64, 128, 9), ("rb2.c2", 16, 32, 128, 128, 9), ("rb2.sc", 16, 32, 64, 128, 1),
              ("rb3.c1", 8, 16, 128, 256, 9), ("rb3.c2", 8, 16, 256, 256, 9), ("rb3.sc", 8, 16, 128, 256, 1),
              ("txt.k3", 1, 128, 256, 128, 3), ("txt.k4", 1, 128, 256, 128, 4), ("txt.k5", 1, 128, 256, 128, 5)]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    print(f"{'layer':8s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    for name, H, W, Cin, Cout, nt in shapes:
        taps = ops.TAPS3X3 if nt == 9 else (ops.TAPS1X1 if nt == 1 and H > 1 else [(0, j) for j in range(nt)])
        x = torch.randn(B, H, W, Cin, device=DEV)
        dy = torch.randn(B, H, W, Cout, device=DEV)
        w = torch.randn(Cout, Cin, nt, device=DEV) * 0.05
        wf, wd = torch.empty_like(w).flatten(), torch.empty_like(w).flatten()
        ops.pack_weights(w, wf, wd)
        y = torch.empty(B, H, W, Cout, device=DEV)
        dx = torch.empty(B, H, W, Cin, device=DEV)
        rows, _ = ops.conv_stats_rows(B, H, W, Cout)
        stats = torch.empty(rows * 2 * Cout, device=DEV)
        fl = 2.0 * B * H * W * Cin * Cout * nt
        t_f = timeit(lambda: ops.conv_fwd(x, wf, y, stats if H > 1 else None, B, H, W, Cin, Cout, taps))
        line = f"{name:8s} {fl / 1e9:7.2f} | {t_f:8.1f} {fl / t_f / 1e6:6.1f} |"
        tot["fwd"] += t_f
        if H > 1:
            t_d = timeit(lambda: ops.conv_fwd(dy, wd, dx, None, B, H, W, Cout, Cin, taps, negate=True))
            dW = torch.empty(Cout, Cin, nt, device=DEV)
            wk = torch.empty(ops.conv_wgrad_workspace(B, H, W, Cin, Cout, nt), device=DEV)
            t_w = timeit(lambda: ops.conv_wgrad(x, dy, dW, wk, B, H, W, Cin, Cout, taps))
            line += f" {t_d:8.1f} {fl / t_d / 1e6:6.1f} | {t_w:8.1f} {fl / t_w / 1e6:6.1f}"
            tot["dgrad"] += t_d
            tot["wgrad"] += t_w
            totf += 3 * fl
        else:
            totf += fl
        print(line)
    allt = sum(tot.values())
    print(f"total: fwd {tot['fwd']:.0f} us, dgrad {tot['dgrad']:.0f} us, wgrad {tot['wgrad']:.0f} us = {allt:.0f} us; "
          f"{totf / allt / 1e6:.1f} TFLOP/s = {totf / allt / 1e6 / PEAK * 100:.1f}% of fp32 matrix peak")


if __name__ == "__main__":
    main()
