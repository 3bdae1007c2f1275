"""How far do CPU runs of the SAME algorithm drift from the recorded reference loss curve when only
rounding-level details change?  Runs the CPU oracle (oracle/train_ref.py: same torch CPU ops as the reference,
AdamW/clip arithmetic written out by hand) for 100 steps on the recorded batches and modality-dropout decisions,
once as is and once with oneDNN convolutions disabled, and stores the per-step |loss - recorded loss|.

    python tests/golden/loss_curve_sensitivity.py      # ~2 minutes on 8 cores; needs no reference checkout
"""
import contextlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import batch_tensors, seeded_model  # noqa: E402
from oracle import train_ref  # noqa: E402


def run(disable_onednn: bool):
    ref = json.load(open(os.path.join(HERE, "loss_curves.json")))["both"]
    st = train_ref.TrainState(seeded_model("both").state_dict(), "both")
    ctx = torch.backends.mkldnn.flags(enabled=False) if disable_onednn else contextlib.nullcontext()
    dev = []
    with ctx:
        for s in range(100):
            b = batch_tensors(32 * s, 32)
            da, dt = ref["drops"][s]
            r = st.step(b["audio"], b["input_ids"], b["label"], da, dt)
            dev.append(abs(r["loss"] - ref["loss"][s]))
    return dev


if __name__ == "__main__":
    out = {"what": "per-step |CPU oracle loss - recorded reference loss|, 100 steps, B=32, seed 42",
           "deviation": {"oracle_cpu_fp32": run(False), "oracle_cpu_fp32_onednn_off": run(True)}}
    with open(os.path.join(HERE, "loss_curve_sensitivity.json"), "w") as f:
        json.dump(out, f)
    for k, v in out["deviation"].items():
        print(k, "max", max(v), "first5", max(v[:5]))
