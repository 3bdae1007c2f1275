#!/usr/bin/env python3
"""Host-side cost of replaying the captured training step: time spent inside graph.replay() (enqueue only) against
the GPU time of a step.  If the two are close the step is bound by the host's enqueue, not by the kernels."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sentiment-analysis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import batch_tensors, seeded_model  # noqa: E402
from mmsa_amd.frontend import LogMelFrontEnd  # noqa: E402
from mmsa_amd.trainer import TrainStep  # noqa: E402


def main():
    dev = "cuda"
    m = seeded_model("both").to(dev).train()
    ts = TrainStep(m, frontend=LogMelFrontEnd(dev), use_graph=True)
    b = batch_tensors(0, 32, device=dev, wave=True)
    kw = dict(wave=b["wave"], input_ids=b["input_ids"], labels=b["label"])
    for _ in range(10):
        ts.step(**kw)
    torch.cuda.synchronize()
    g = ts._graph[0]
    n = 200
    # (a) enqueue cost with an idle GPU between launches
    host = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    # (b) back-to-back replays
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # (c) one replay, GPU time from events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"replay() host time, idle GPU: median {sorted(host)[len(host) // 2] * 1e6:.0f} us")
    print(f"{n} back-to-back replays: enqueue returned after {t_enq / n * 1e6:.0f} us/step, "
          f"all finished after {t_all / n * 1e6:.0f} us/step")
    print(f"single replay, GPU time between events: {e0.elapsed_time(e1) * 1e3:.0f} us")


if __name__ == "__main__":
    main()
