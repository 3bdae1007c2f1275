"""Developer aid (not a pytest file): per-tensor gradient error of the audio model on a 64 x 460 spectrogram against
the float64 oracle, for the fp32 CPU oracle and for the HIP path under each MMSA_IGEMM3 mode (argument: seeds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import SMALL_VOCAB, seeded_model  # noqa: E402
from oracle import model_ref  # noqa: E402

DEV = "cuda"


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [5]
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)
        audio = torch.randn(2, 1, 64, 460, generator=g)
        label = torch.tensor([0, 2])
        rows = {}
        for mode in ("0", "2"):
            os.environ["MMSA_IGEMM3"] = mode
            m = seeded_model("audio", vocab=SMALL_VOCAB).to(DEV).train()
            sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            out = m(audio=audio.to(DEV), labels=label.to(DEV))
            out["loss"].backward()
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            _, g64, _ = model_ref.forward_backward(sd64, "audio", audio.double(), None, label)
            _, g32, _ = model_ref.forward_backward(sd, "audio", audio, None, label)
            for k, p in m.named_parameters():
                r = g64[k]
                n = float(torch.linalg.vector_norm(r)) + 1e-30
                e = float(torch.linalg.vector_norm(p.grad.detach().cpu().double() - r)) / n
                c = float(torch.linalg.vector_norm(g32[k].double() - r)) / n
                rows.setdefault(k, {})[mode] = e
                rows[k]["cpu32"] = c
        print(f"== seed {seed}: relative L2 error vs float64")
        for k, v in rows.items():
            if max(v.values()) > 2e-4:
                print(f"   {k:45s} general {v['0']:.2e}  rows-of-three {v['2']:.2e}  cpu fp32 {v['cpu32']:.2e}")


if __name__ == "__main__":
    main()
