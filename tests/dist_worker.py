"""Helper process of test_zy_two_ranks.py::test_two_ranks_sharing_one_gpu: one data-parallel rank.

usage: python dist_worker.py RANK WORLD PORT OUT_PREFIX USE_GRAPH
Every rank uses cuda:0 (the test box has one GPU; RCCL wants one GPU per rank) and the gloo backend, for which
mmsa_amd.parallel stages the gradient buckets through pinned host memory; what is under test is the host logic of the
multi-rank step (split backward, exchanges issued early and waited for late, clip on the reduced gradient, the
chain of graphs), not the transport."""
import os
import sys

# two processes time-share one GPU's hardware queues here: with four streams each, cross-stream waits between
# descheduled queues make progress crawl, so every rank keeps its step on one stream
os.environ["MMSA_FORK"] = "0"
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", os.environ.get("MMSA_TEST_PACKET_CAPTURE", "0"))

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import SMALL_VOCAB, batch_tensors, seeded_model  # noqa: E402


def main():
    import faulthandler
    rank, world, port, prefix, use_graph = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1"
    trace = open(f"{prefix}.rank{rank}.trace", "w")
    faulthandler.dump_traceback_later(45, exit=True, file=trace)      # a stuck rank reports where and exits
    # one gloo worker thread: collectives run strictly in issue order, as they do on an RCCL stream
    opts = dist.ProcessGroupGloo._Options()
    opts._devices = [dist.ProcessGroupGloo.create_default_device()]
    opts._threads = 1
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            pg_options=opts)
    from mmsa_amd.trainer import TrainStep
    m = seeded_model("both", vocab=SMALL_VOCAB).to("cuda").train()
    m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    ts = TrainStep(m, use_graph=use_graph)
    assert ts.dist and ts.world == world and ts.overlap == (os.environ.get("MMSA_OVERLAP_ALLREDUCE", "auto") != "0")
    losses = []
    if os.environ.get("MMSA_DIST_DEBUG"):
        import threading
        import time
        marks = {}

        def wrap(name):
            orig = getattr(ts, name)

            def f(*a):
                ev = torch.cuda.Event()
                ev.record()
                marks[name] = ev
                print(f"rank {rank} enter {name}", file=trace, flush=True)
                out = orig(*a)
                print(f"rank {rank} leave {name}", file=trace, flush=True)
                return out
            setattr(ts, name, f)
        wrap("_allreduce_early")
        wrap("_allreduce_late")

        def watchdog():
            while True:
                time.sleep(10)
                print(f"rank {rank} watchdog: " + ", ".join(f"{k} gpu-reached={v.query()}" for k, v in marks.items()),
                      file=trace, flush=True)
        threading.Thread(target=watchdog, daemon=True).start()
    for s in range(6):
        b = batch_tensors(100 * s + 8 * rank, 8, vocab=SMALL_VOCAB, device="cuda")
        losses.append(float(ts.step(audio=b["audio"], input_ids=b["input_ids"], labels=b["label"])))
        gn = float(ts.grad_norm())
        assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e3 and gn == gn and 0.0 < gn < 1e6, (s, losses[-1], gn)
        print(f"rank {rank} step {s} loss {losses[-1]:.5f}", file=trace, flush=True)
    torch.cuda.synchronize()
    # the row-sparse exchange alone, on GPU tensors: bit-equal to the dense sum of the same two tables
    from mmsa_amd import parallel
    gen = torch.Generator().manual_seed(7 + rank)
    ids = torch.randint(0, SMALL_VOCAB, (96,), generator=gen)
    ids[5] = ids[40]                                              # a repeated id
    table = torch.zeros(SMALL_VOCAB, 16)
    table[ids] = torch.randn(96, 16, generator=gen)
    table, ids = table.cuda(), ids.cuda()
    dense = table.clone()
    for h in parallel.allreduce_buckets(dense.view(-1), [(0, dense.numel())], None, async_op=True):
        h.wait()
    ex = parallel.SparseRowExchange(96, 16, SMALL_VOCAB, torch.device("cuda"))
    ex.pack(table, ids)
    ex.finish(table, ex.start())
    torch.cuda.synchronize()
    exact = bool(torch.equal(table, dense))
    # sharded inference (multimodal_test.py --modality both over several GPUs): each rank scores its contiguous shard
    # of a 21-utterance test set (ragged: 11 + 10) on the GPU, the predictions are gathered back into dataset order on
    # every rank, and must equal what one process gets for the whole set with the same (trained, replicated) weights
    # BatchNorm's running statistics are per rank (each rank normalises its own batch): every replica takes rank 0's,
    # the ones a checkpoint written by rank 0 would hold
    bn_before = torch.cat([b.detach().flatten().float().cpu() for b in m.buffers()])
    parallel.broadcast_buffers(m)
    bn_after = torch.cat([b.detach().flatten().float().cpu() for b in m.buffers()])
    assert (rank == 0) == bool(torch.equal(bn_before, bn_after)), "the ranks' running statistics should have differed"
    m.eval()
    n_test = 21
    test = batch_tensors(1000, n_test, vocab=SMALL_VOCAB, device="cuda")
    with torch.no_grad():
        whole = m(audio=test["audio"], input_ids=test["input_ids"])
        lo, hi = parallel.shard_range(n_test, rank, world)
        mine = m(audio=test["audio"][lo:hi], input_ids=test["input_ids"][lo:hi])
    probs, preds = mine["probabilities"], mine["probabilities"].argmax(1)
    gp, gr = parallel.gather_predictions(probs, preds, n_test)
    torch.cuda.synchronize()
    infer_err = float((gp - whole["probabilities"]).abs().max())
    infer_ok = bool(gp.shape == (n_test, 3) and gp.is_cuda and torch.equal(gr, whole["probabilities"].argmax(1)))
    torch.save({"flat": m._arena.flat.cpu(), "losses": losses, "sparse": bool(ts._use_sparse),
                "exchange_exact": exact, "infer_err": infer_err, "infer_ok": infer_ok, "infer_probs": gp.cpu(), "buffers": bn_after},
               f"{prefix}.rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
