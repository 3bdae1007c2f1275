#!/usr/bin/env python3
"""Isolated timing of the fused fusion-head launches (developer aid):  python tools/head_bench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sentiment-analysis_amd"))
import torch  # noqa: E402

from mmsa_amd import head, ops  # noqa: E402
from mmsa_amd.model import MultimodalSentimentModel  # noqa: E402
from mmsa_amd.workspace import Grads, Workspace  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.manual_seed(0)
    m = MultimodalSentimentModel("both", vocab_size=1000, audio_dropout=0.0, text_dropout=0.0, fusion_dropout=0.0).cuda()
    m._ensure_arena()
    ws = Workspace("cuda", persistent=True)
    a, t = torch.randn(B, 256, device="cuda"), torch.randn(B, 384, device="cuda")
    lab = torch.randint(0, 3, (B,), device="cuda")
    grads = Grads()
    st = head.head_forward_both(m, a, t, lab, True, None, None, 0.0, 0, True, ws)
    tab_f = ops.head_both_table(st.tensors)
    head.head_backward_both(m, st, st.dlogits, grads, ws)
    tab_b = ops.head_both_table(st.tensors)
    print("fwd  %.1f us" % timeit(lambda: ops.head_both("fwd", tab_f, st.dims, 0.0, 0)))
    print("bwd  %.1f us" % timeit(lambda: ops.head_both("bwd", tab_b, st.dims, 0.0, 0)))
    print("wgrd %.1f us" % timeit(lambda: ops.head_both("bwd_params", tab_b, st.dims)))


if __name__ == "__main__":
    main()
