#!/usr/bin/env python3
"""Noise gate alone, for profiling: python tools/denoise_bench.py [batch] [samples]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-sentiment-analysis_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from mmsa_amd.frontend import NoiseReducer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 66150
nr = NoiseReducer(22050, "cuda")
wave = torch.randn(B, n, device="cuda") * 0.1
out = torch.empty_like(wave)
for _ in range(3):
    nr(wave, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    nr(wave, out=out)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / 20
print(f"B={B} n={n}: {t * 1e3:.1f} us per batch, {B / t * 1e3:.0f} utt/s")
