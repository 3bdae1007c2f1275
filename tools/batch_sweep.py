#!/usr/bin/env python3
"""Training-step throughput against the per-GPU batch size (the headline metric is defined at 32): shows how much of
the distance to the matrix peak is problem size.  Prints one JSON object.

    python tools/batch_sweep.py [batch ...]
"""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-sentiment-analysis_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from mmsa_amd import data  # noqa: E402
from mmsa_amd.frontend import LogMelFrontEnd  # noqa: E402
from mmsa_amd.model import MultimodalSentimentModel  # noqa: E402
from mmsa_amd.trainer import TrainStep  # noqa: E402

FLOP_PER_UTT = 2430.3e6
out = {}
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    torch.manual_seed(42)
    model = MultimodalSentimentModel("both", audio_dropout=0.5, text_dropout=0.3, fusion_dropout=0.6).to("cuda").train()
    ts = TrainStep(model, frontend=LogMelFrontEnd("cuda"), use_graph=True)
    b = data.make_batch(0, B, with_wave=True)
    dev = {k: torch.from_numpy(v).to("cuda") for k, v in b.items() if k in ("wave", "input_ids", "label")}
    for _ in range(10):
        ts.step(wave=dev["wave"], input_ids=dev["input_ids"], labels=dev["label"])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    s.record()
    for _ in range(n):
        ts.step(wave=dev["wave"], input_ids=dev["input_ids"], labels=dev["label"])
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / n * 1e-3
    out[f"B{B}"] = {"ms_per_step": round(t * 1e3, 3), "utt_per_s": round(B / t), "model_tflops": round(B / t * FLOP_PER_UTT / 1e12, 1),
                    "frac_of_fp32_matrix_peak": round(B / t * FLOP_PER_UTT / 157.3e12, 3)}
    del ts, model
    torch.cuda.empty_cache()
print(json.dumps(out))
