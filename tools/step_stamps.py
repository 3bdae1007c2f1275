#!/usr/bin/env python3
"""When does each stream of the replayed training step really reach its fork / join points?

    python tools/step_stamps.py [replays]          (environment switches of the package apply, e.g. MMSA_WGRAD=old)

Runs bench.py's configuration (batch 32, graph replay) with ops.STAMPS set: one-thread kernels between the launches
record the GPU's 100 MHz real-time counter.  Prints the median over the replays of every stamp, in us since the
step's first stamp, sorted by time.  No profiler is attached, so the concurrency is the timed run's (each stamp adds a
launch of its own, about 2 us on its stream)."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-sentiment-analysis_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from mmsa_amd import data, ops
    from mmsa_amd.frontend import LogMelFrontEnd
    from mmsa_amd.model import MultimodalSentimentModel
    from mmsa_amd.trainer import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    model = MultimodalSentimentModel("both", audio_dropout=0.5, text_dropout=0.3, fusion_dropout=0.6).to(dev).train()
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    slots = {}
    ops.STAMPS = (buf, slots)
    ts = TrainStep(model, lr=1e-3, weight_decay=1e-4, max_norm=1.0, frontend=LogMelFrontEnd(dev), use_graph=True)
    b = data.make_batch(0, 32, with_wave=True)
    b = {k: torch.from_numpy(b[k]).to(dev) for k in ("wave", "input_ids", "label")}
    rows = []
    for i in range(replays + 5):
        ts.step(wave=b["wave"], input_ids=b["input_ids"], labels=b["label"])
        torch.cuda.synchronize()
        if i >= 5:
            rows.append(buf.cpu().clone())
    t = torch.stack(rows)[:, :len(slots)].double()
    t = (t - t[:, slots["step.start"]:slots["step.start"] + 1]) / 100.0     # 10 ns ticks -> us
    med = t.median(dim=0).values
    for label, slot in sorted(slots.items(), key=lambda kv: float(med[kv[1]])):
        print(f"{float(med[slot]):9.1f} us  {label}")


if __name__ == "__main__":
    main()
