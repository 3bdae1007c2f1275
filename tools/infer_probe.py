#!/usr/bin/env python3
"""Inference of 1024 utterances (front end + eval forward of the `both` model), a few times: the program to put
behind `rocprofv3 --kernel-trace --stats` for a per-kernel breakdown of the inference path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sentiment-analysis_amd"))
import torch
from mmsa_amd import data
from mmsa_amd.frontend import LogMelFrontEnd
from mmsa_amd.model import MultimodalSentimentModel

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    dev = "cuda"
    torch.manual_seed(42)
    model = MultimodalSentimentModel("both").to(dev).eval()
    fe = LogMelFrontEnd(dev)
    b = data.make_batch(0, 64, with_wave=True)
    wave = torch.from_numpy(b["wave"]).to(dev).repeat(B // 64, 1)
    ids = torch.from_numpy(b["input_ids"]).to(dev).repeat(B // 64, 1)
    def infer():
        with torch.no_grad():
            return model(audio=fe(wave), input_ids=ids)["logits"].argmax(1)
    for _ in range(2): infer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): infer()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"B={B}: {t*1e3:.3f} ms per batch, {B/t:.0f} utt/s, {B*811711872/t/1e12:.1f} TFLOP/s")

if __name__ == "__main__":
    main()
