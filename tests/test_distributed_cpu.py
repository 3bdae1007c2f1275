"""Multi-process gloo tests (CPU; world sizes 2, 3, 4 and 8) of the data-parallel host logic: sharding, bucketed
gradient all-reduce (the backend's and the explicit reduce-scatter + all-gather), the row-sparse exchange of the
embedding gradient, the gather of ragged prediction shards, the broadcast of BatchNorm buffers.  The per-rank
gradients come from the CPU oracle (text modality: no BatchNorm, so the mean of per-rank gradients must equal the
single-process gradient of the concatenated batch)."""
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


def _rank_order_sum(t):
    """sum over ranks of ``t`` taken in rank order (t_0 + t_1) + t_2 ...: what a deterministic exchange must equal."""
    world = dist.get_world_size()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    acc = parts[0].clone()
    for p in parts[1:]:
        acc += p
    return acc


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mmsa_amd import parallel
        from oracle import model_ref
        sd = seeded_model("text", vocab=SMALL_VOCAB).state_dict()
        keys = model_ref.param_keys(sd)
        gb = world * max(2, 8 // world)
        full = batch_tensors(0, gb, vocab=SMALL_VOCAB)
        # rank r takes rows r::world of the global batch, as train_batches deals them (equal shares: drop_last)
        idx = torch.arange(gb)[rank::world]
        out, grads, _ = model_ref.forward_backward(sd, "text", None, full["input_ids"][idx], full["label"][idx])
        mine = torch.cat([grads[k].flatten() for k in keys]) * (len(idx) / gb)
        flat = mine.clone()
        buckets = parallel.make_buckets(flat.numel(), 100_000)
        assert len(buckets) > 3 and buckets[0] == (0, 100_000) and buckets[-1][1] == flat.numel()
        handles = parallel.allreduce_buckets(flat, buckets, async_op=True)
        for h in handles:
            h.wait()
        _, ref_g, _ = model_ref.forward_backward(sd, "text", None, full["input_ids"], full["label"])
        ref = torch.cat([ref_g[k].flatten() for k in keys])
        err = float((flat - ref).abs().max() / ref.abs().max())
        # sharded inference bookkeeping: ragged shards (the last ranks hold fewer utterances, or none)
        ok = True
        for n in (11, 2610):
            lo, hi = parallel.shard_range(n, rank, world)
            probs = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
            preds = torch.arange(lo, hi)
            P, R = parallel.gather_predictions(probs, preds, n)
            ok = ok and torch.equal(R, torch.arange(n)) and torch.equal(P[:, 0], torch.arange(n, dtype=torch.float32))
        tb = parallel.train_batches(200, 4, 7, rank, world)
        # the explicit reduce-scatter + all-gather: bucket lengths that the world size does not divide (the remainder
        # goes through the backend), sums in rank order: identical on all ranks, equal to the rank-order sum bit for
        # bit, and at two ranks to the backend's all-reduce as well
        direct = mine.clone()
        db = parallel.make_buckets(direct.numel(), 4096 * world + 1)
        assert any((hi - lo) % world for lo, hi in db) or world == 1
        for h in parallel.allreduce_buckets(direct, db, async_op=True, algo="direct"):
            h.wait()
        direct_ok = True
        for lo, hi in db:
            cut = hi - (hi - lo) % world
            direct_ok = direct_ok and torch.equal(direct[lo:cut], _rank_order_sum(mine[lo:cut]))
        direct_ok = direct_ok and float((direct - flat).abs().max()) <= 1e-6 * float(flat.abs().max())
        if world == 2:
            direct_ok = direct_ok and torch.equal(direct, flat)
        parallel.assert_same_on_all_ranks(direct.view(torch.uint8), "directly summed gradient")
        sync = torch.zeros(world + 1)
        parallel.allreduce_buckets(sync.add_(1.0), [(0, world + 1)], algo="direct")    # blocking form, ragged bucket
        direct_ok = direct_ok and bool((sync == world).all())
        # row-sparse exchange of the embedding table's gradient against the dense sum of the same tables in rank order
        table = grads["text_model.embedding.weight"].clone()
        dense = _rank_order_sum(table)
        # one extra id outside the table: it travels as a zero row
        ids = torch.cat([full["input_ids"][idx].reshape(-1), torch.tensor([SMALL_VOCAB + 5])])
        ex = parallel.SparseRowExchange(ids.numel(), table.shape[1], table.shape[0], torch.device("cpu"))
        ex.pack(table, ids)
        ex.finish(table, ex.start())
        touched = torch.zeros(table.shape[0], dtype=torch.bool)
        touched[ex.ids_all.reshape(-1)] = True
        sparse_ok = (torch.equal(table, dense) and not table[~touched].any().item()
                     and int(touched.sum()) < table.shape[0])
        parallel.assert_same_on_all_ranks(table.view(torch.uint8), "summed embedding gradient")
        # a table WITHOUT a zero row 0 (no padding row), ids outside the table ahead of a genuine id 0
        g = torch.Generator().manual_seed(100 + rank)
        t2 = torch.zeros(40, 8)
        ids2 = torch.tensor([47, -3, 0, 5, 0, 39, 47, 5])
        valid = ids2[(ids2 >= 0) & (ids2 < 40)].unique()
        t2[valid] = torch.randn(len(valid), 8, generator=g)
        want2 = _rank_order_sum(t2)
        ex2 = parallel.SparseRowExchange(ids2.numel(), 8, 40, torch.device("cpu"))
        ex2.pack(t2, ids2)
        ex2.finish(t2, ex2.start())
        sparse_ok = sparse_ok and torch.equal(t2, want2) and bool(t2[0].any())
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


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_allreduce_sharding_and_exchanges_match_single_process(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    for r in range(world):
        err, ok, _ = ret[r]
        assert err < 1e-5, err
        assert all(ok), ok
    per_rank = [ret[r][2] for r in range(world)]
    gb = 4 * world
    assert all(len(b) == 200 // gb for b in per_rank)         # drop_last over global batches
    seen = [i for b in per_rank for batch in b for i in batch]
    assert len(seen) == len(set(seen)) == 200 // gb * gb      # disjoint shards, nothing repeated
    assert all(len(batch) == 4 for b in per_rank for batch in b)


def test_single_process_helpers():
    from mmsa_amd import parallel
    assert parallel.world_info() == (0, 1)
    assert parallel.shard_range(2610, 7, 8) == (2289, 2610) and parallel.shard_range(3, 5, 8) == (3, 3)
    assert parallel.make_buckets(10, 4) == [(0, 4), (4, 8), (8, 10)]
    p, r = torch.ones(3, 3), torch.zeros(3, dtype=torch.long)
    assert parallel.gather_predictions(p, r, 3)[0] is p
