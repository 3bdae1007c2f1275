"""Two data-parallel ranks on one GPU (collected late: the file name sorts behind the single-process GPU tests).

The worker processes (tests/dist_worker.py) run the multi-rank step of mmsa_amd.trainer.TrainStep -- split backward,
asynchronous gradient exchanges, clip on the reduced gradient, and in graph mode the chain of three graphs with the
collectives between them -- with gloo over GPU tensors, because a test box has one GPU and RCCL wants one GPU per rank.
A stuck rank reports where it waits and exits non-zero (faulthandler), the parent gives up after 150 s."""
import os

import pytest
import torch

from conftest import SMALL_VOCAB, batch_tensors, model_inputs, seeded_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run_two_ranks(use_graph, prefix, port_base, **env):
    import subprocess
    import sys
    port = str(port_base + (os.getpid() % 150))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, prefix, use_graph],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env={**os.environ, **env})
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    traces = ""
    for r in range(2):
        path = f"{prefix}.rank{r}.trace"
        if os.path.exists(path):
            traces += f"--- rank {r}\n" + open(path).read()[-3000:]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs) + traces
    r0 = torch.load(prefix + ".rank0.pt", weights_only=True)
    r1 = torch.load(prefix + ".rank1.pt", weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"])
    return r0, r1


def test_row_sparse_and_dense_embedding_exchange_agree(tmp_path):
    """Six steps with the embedding table's gradient exchanged as (ids, rows) against six steps with the dense
    all-reduce.  The exchange is bit-exact (the workers also check that on a seeded table, see dist_worker.py) and the
    step holds no order-dependent sum, so the two runs end in the same bits."""
    sp, _ = _run_two_ranks("0", str(tmp_path / "sp"), 29400, MMSA_SPARSE_EMB="1")
    de, _ = _run_two_ranks("0", str(tmp_path / "de"), 29600, MMSA_SPARSE_EMB="0")
    assert sp["sparse"] is True and de["sparse"] is False
    assert sp["exchange_exact"] and de["exchange_exact"]
    assert sp["losses"] == de["losses"], (sp["losses"], de["losses"])
    assert torch.equal(sp["flat"], de["flat"])


def test_two_ranks_sharing_one_gpu(tmp_path):
    """Two data-parallel ranks (gloo over GPU tensors, both on this GPU), launched eagerly and as the chain of three
    graphs: the two modes end in the same bits, replicas stay bit-identical to each other, and both follow a
    single-process emulation: per step the gradient is the mean of the two ranks' autograd gradients (each already
    holds the regulariser's gradient once), clipped at 1.0 and applied by torch.optim.AdamW.  The embedding table's
    gradient travels row-sparse here (MMSA_SPARSE_EMB=1), so both exchanges are under test."""
    r0, r1 = _run_two_ranks("0", str(tmp_path / "dp"), 29800, MMSA_SPARSE_EMB="1")
    g0, g1 = _run_two_ranks("1", str(tmp_path / "dpg"), 30000, MMSA_SPARSE_EMB="1")
    assert r0["sparse"] is True and g0["sparse"] is True
    assert r0["losses"] == g0["losses"] and r1["losses"] == g1["losses"], (r0["losses"], g0["losses"])
    assert torch.equal(r0["flat"], g0["flat"])
    # sharded inference behind the training steps (see dist_worker.py): gathered predictions in dataset order, equal on
    # both ranks and equal to one process scoring the whole set (per-utterance results do not depend on the batch
    # they are scored in beyond rounding of the tiled reductions)
    for a, b in ((r0, r1), (g0, g1)):
        assert torch.equal(a["buffers"], b["buffers"])          # running statistics broadcast from rank 0
        assert a["infer_ok"] and b["infer_ok"] and torch.equal(a["infer_probs"], b["infer_probs"])
        assert max(a["infer_err"], b["infer_err"]) < 1e-5, (a["infer_err"], b["infer_err"])

    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
    params = list(m.parameters())
    for s in range(6):
        acc, ls = None, []
        for rank in range(2):
            b = batch_tensors(100 * s + 8 * rank, 8, vocab=SMALL_VOCAB, device=DEV)
            opt.zero_grad()
            out = m(**model_inputs("both", b), labels=b["label"])
            out["loss"].backward()
            ls.append(float(out["loss"].detach()))
            g = [p.grad.detach().clone() for p in params]
            acc = g if acc is None else [a + x for a, x in zip(acc, g)]
        for p, a in zip(params, acc):
            p.grad = a / 2
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        # The emulation is another fp32 evaluation of the same step (torch's AdamW and clip arithmetic, the
        # regulariser's gradient through autograd).  The first steps pin the mechanics (mean of the ranks' gradients,
        # clip on the reduced gradient, AdamW): equal to rounding.  From there on two fp32 evaluations drift apart by a
        # factor of 5-25 per step (measured: 0, 2e-7, 7e-6, 1.7e-4, 8e-4, ...; Adam turns noise-level gradient
        # differences into +-lr steps), so the later steps are only held to a bound on that drift.
        bar = 2e-5 if s < 3 else 5e-3
        dev = max(abs(ls[0] - r0["losses"][s]), abs(ls[1] - r1["losses"][s]))
        assert dev < bar, (s, dev, ls, r0["losses"][s], r1["losses"][s])
    got = m._arena.views_of(r0["flat"].to(DEV))
    for (k, pa), pb in zip(m.named_parameters(), got):
        if "conv1.0.bias" in k:
            continue
        d = float(torch.linalg.vector_norm(pa.detach() - pb))
        assert d < 0.05 * 6e-3 * pa.numel() ** 0.5 + 1e-6, k
