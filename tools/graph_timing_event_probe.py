#!/usr/bin/env python3
"""Do timing events recorded as nodes of a captured graph ("external" events) return kernel durations on replay?

Captures   e0 | matmul | e1 | (side stream) matmul | e2   and replays it three times, printing the elapsed times."""
import os
import sys
import torch

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
a = torch.randn(4096, 4096, device="cuda")
b = torch.empty_like(a)
c = torch.empty_like(a)
torch.mm(a, a, out=b)
torch.cuda.synchronize()
try:
    evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(4)]
except TypeError as exc:
    print("no external events in this torch:", exc)
    sys.exit(0)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    evs[0].record()
    torch.mm(a, a, out=b)
    evs[1].record()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        evs[2].record()
        torch.mm(a, a, out=c)
        evs[3].record()
    torch.mm(b, a, out=b)
    cur.wait_stream(side)
for it in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", it, "main mm %.3f ms, side mm %.3f ms" % (evs[0].elapsed_time(evs[1]), evs[2].elapsed_time(evs[3])),
          flush=True)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); torch.mm(a, a, out=b); e.record(); torch.cuda.synchronize()
print("eager mm %.3f ms" % s.elapsed_time(e))
