"""World-size-2 gloo tests (CPU) of the data-parallel host logic: sharding, bucketed gradient all-reduce, and the
gather of sharded predictions.  The per-rank gradients come from the CPU oracle (text modality: no BatchNorm,
so the mean of per-rank gradients must equal the single-process gradient of the concatenated batch)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import SMALL_VOCAB, batch_tensors, seeded_model


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mmsa_amd import parallel
        from oracle import model_ref
        sd = seeded_model("text", vocab=SMALL_VOCAB).state_dict()
        keys = model_ref.param_keys(sd)
        full = batch_tensors(0, 8, vocab=SMALL_VOCAB)
        # rank r takes rows r::world of the global batch, as train_batches deals them
        idx = torch.arange(8)[rank::world]
        out, grads, _ = model_ref.forward_backward(sd, "text", None, full["input_ids"][idx], full["label"][idx])
        flat = torch.cat([grads[k].flatten() for k in keys]) / world     # per-rank mean -> global mean after SUM
        buckets = parallel.make_buckets(flat.numel(), 100_000)
        assert len(buckets) > 3 and buckets[0] == (0, 100_000) and buckets[-1][1] == flat.numel()
        handles = parallel.allreduce_buckets(flat, buckets, async_op=True)
        for h in handles:
            h.wait()
        _, ref_g, _ = model_ref.forward_backward(sd, "text", None, full["input_ids"], full["label"])
        ref = torch.cat([ref_g[k].flatten() for k in keys])
        err = float((flat - ref).abs().max() / ref.abs().max())
        # sharded inference bookkeeping
        n = 11
        lo, hi = parallel.shard_range(n, rank, world)
        probs = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
        preds = torch.arange(lo, hi)
        P, R = parallel.gather_predictions(probs, preds, n)
        ok = torch.equal(R, torch.arange(n)) and torch.equal(P[:, 0], torch.arange(n, dtype=torch.float32))
        tb = parallel.train_batches(50, 4, 7, rank, world)
        # the explicit reduce-scatter + all-gather against the backend's all-reduce (rank-order sums: same bits)
        direct = torch.cat([grads[k].flatten() for k in keys]) / world
        direct = torch.cat([direct, torch.zeros((-direct.numel()) % (world * 4))])
        db = parallel.make_buckets(direct.numel(), 4096 * world)
        for h in parallel.allreduce_buckets(direct, db, async_op=True, algo="direct"):
            h.wait()
        direct_ok = torch.equal(direct[:flat.numel()], flat)
        try:
            parallel.allreduce_buckets(torch.zeros(world + 1), [(0, world + 1)], algo="direct")
            direct_ok = False
        except ValueError:
            pass
        # row-sparse exchange of the embedding table's gradient against the dense all-reduce of the same table
        table = grads["text_model.embedding.weight"].clone()
        dense = table.clone()
        dist.all_reduce(dense)
        # one extra id outside the table: it travels as a zero row
        ids = torch.cat([full["input_ids"][idx].reshape(-1), torch.tensor([SMALL_VOCAB + 5])])
        ex = parallel.SparseRowExchange(ids.numel(), table.shape[1], table.shape[0], torch.device("cpu"))
        ex.pack(table, ids)
        ex.finish(table, ex.start())
        touched = torch.zeros(table.shape[0], dtype=torch.bool)
        touched[ex.ids_all.reshape(-1)] = True
        sparse_ok = (torch.equal(table, dense) and not table[~touched].any().item()
                     and int(touched.sum()) < table.shape[0])
        parallel.assert_same_on_all_ranks(table, "summed embedding gradient")
        try:
            parallel.assert_same_on_all_ranks(torch.full((4,), float(rank)), "rank-dependent tensor")
            differs = False
        except RuntimeError:
            differs = True
        # BatchNorm running statistics are per rank: broadcast_buffers gives every replica rank 0's
        bn = torch.nn.BatchNorm1d(4)
        bn.running_mean.fill_(float(rank + 1))
        bn.num_batches_tracked.fill_(10 * (rank + 1))
        parallel.broadcast_buffers(bn)
        buffers_ok = (bool((bn.running_mean == 1.0).all()) and int(bn.num_batches_tracked) == 10
                      and bool((bn.running_var == 1.0).all()))
        ret[rank] = (err, (ok, sparse_ok, differs, direct_ok, buffers_ok), [b.tolist() for b in tb])
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    for r in range(world):
        err, ok, _ = ret[r]
        assert err < 1e-5, err
        assert all(ok), ok
    b0, b1 = ret[0][2], ret[1][2]
    assert len(b0) == len(b1) == 50 // 8                      # drop_last over global batches of 8
    seen = [i for a, b in zip(b0, b1) for i in a + b]
    assert len(seen) == len(set(seen)) == 48                  # disjoint shards, nothing repeated


def test_single_process_helpers():
    from mmsa_amd import parallel
    assert parallel.world_info() == (0, 1)
    assert parallel.shard_range(2610, 7, 8) == (2289, 2610) and parallel.shard_range(3, 5, 8) == (3, 3)
    assert parallel.make_buckets(10, 4) == [(0, 4), (4, 8), (8, 10)]
    p, r = torch.ones(3, 3), torch.zeros(3, dtype=torch.long)
    assert parallel.gather_predictions(p, r, 3)[0] is p
