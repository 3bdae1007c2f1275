#!/usr/bin/env python3
"""Stage timeline inside the fused head's forward launch (developer aid; needs a library whose head.hip was built with
-DMMSA_LAB_HEAD_STAMPS):  python tools/head_stamps.py [B]
Prints, for workgroup 0, the time between stage boundaries with the weights warm in the L2 and after a 1 GiB fill."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sentiment-analysis_amd"))
import torch  # noqa: E402

from mmsa_amd import head, ops  # noqa: E402
from mmsa_amd.model import MultimodalSentimentModel  # noqa: E402
from mmsa_amd.workspace import Grads, Workspace  # noqa: E402

BWD_STAGES = ["start->staged/loaded", "c2 matvec^T", "LN c1 bwd", "c1 matvec^T", "LN c0 bwd", "c0 matvec^T",
              "softmax bwd", "attention matvec^T", "LN proj A bwd", "proj A matvec^T", "LN proj T bwd", "proj T matvec^T",
              "feature LN bwd (audio)", "feature LN bwd (text)"]
STAGES = ["start->staged/loaded", "feature LN", "proj A (+T) matvec", "LN proj A", "proj T matvec + LN", "attention",
          "softmax+fuse", "c0 matvec", "LN c0", "c1 matvec", "LN c1", "c2 matvec", "logits+loss grad"]


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
    head.head_backward_both(m, st, st.dlogits, grads, ws)
    tab = ops.head_both_table(st.tensors)
    buf = st.tensors["dwpa"]
    assert buf is not None and buf.numel() == 256 * 256, "slot numbering changed"
    big = torch.empty(1 << 28, device="cuda")
    for which, names in (("fwd", STAGES), ("bwd", BWD_STAGES)):
        for label, flush in (("warm", False), ("after 1 GiB fill", True)):
            rows, mhz = [], []
            for _ in range(5):
                if flush:
                    big.fill_(1.0)
                torch.cuda.synchronize()
                ops.head_both(which, tab, st.dims, 0.0, 0)
                torch.cuda.synchronize()
                s = buf.reshape(-1).view(torch.int64)[:64].cpu().tolist()
                n = len(names)
                rows.append([(s[i + 1] - s[i]) * 10 for i in range(n)])
                mhz.append((s[32 + n] - s[32]) / max(1, (s[n] - s[0]) * 10) * 1e3)
            print(which, label, "(ns per stage, five launches; shader clock counter ran at %s MHz)"
                  % " ".join("%.0f" % v for v in mhz))
            for i, name in enumerate(names):
                print("  %-24s %s" % (name, " ".join("%6d" % r[i] for r in rows)))
            print("  %-24s %s" % ("total", " ".join("%6d" % sum(r) for r in rows)))


if __name__ == "__main__":
    main()
