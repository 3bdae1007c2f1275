"""Data-parallel host logic (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The reference is single-process; sharding follows its data conventions: training batches are drawn from a seeded
permutation with ``drop_last`` (`CNN implementation/multimodal_train.py:435-442`), test sets are walked in order
(`multimodal_test.py:194-212`).  Utterances are independent, so the only exchange per training step is the sum
of the flat gradient buffer; inference exchanges nothing but the final predictions.
"""
from __future__ import annotations

from typing import List, Tuple

import contextlib
import os

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def assert_same_on_all_ranks(blob: torch.Tensor, what: str, group=None, device=None):
    """Raises on every rank unless the byte tensor ``blob`` is identical on all ranks (compared through a 3-word
    digest: length, sum, position-weighted sum; one small collective, meant for set-up time)."""
    b = blob.flatten().to(torch.int64)
    w = torch.arange(1, b.numel() + 1, dtype=torch.int64) % 65521
    digest = torch.stack([torch.tensor(b.numel()), b.sum(), (b * w).sum()]).to(torch.int64)
    if device is not None:
        digest = digest.to(device)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo.cpu(), hi.cpu()):
        raise RuntimeError(f"data-parallel ranks disagree on {what}")


def make_buckets(n: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) slices of a flat buffer of n elements, each at most bucket_elems long."""
    bucket_elems = max(int(bucket_elems), 1)
    return [(i, min(n, i + bucket_elems)) for i in range(0, n, bucket_elems)]


class _HostStagedSum:
    """Sum of a GPU tensor over a process group whose transport works on host memory (gloo).

    gloo accepts GPU tensors, but its own staging (an event on the compute stream, a high-priority pool stream that
    waits for it and copies to pinned memory, a worker thread that synchronises that stream) was seen never to be
    released when the compute stream's last submission was a HIP graph launch of the training step: the compute
    stream's work did not complete until the process was killed -- in a single process as well, with or without
    asynchronous handles, with the graph runtime's pre-recorded packets on or off.  The same step over RCCL, and
    over gloo with the staging below, runs.  So the staging is done here, on the compute stream itself: copy to
    pinned memory behind the producer, and at ``wait()`` block the host on that copy, reduce the host buffer, copy it
    back in stream order."""

    def __init__(self, dst: torch.Tensor, group):
        self.dst, self.group = dst, group
        self.host = torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
        self.host.copy_(dst, non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()

    def wait(self):
        self.copied.synchronize()
        dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=self.group)
        self.dst.copy_(self.host, non_blocking=True)
        return True


class _DirectSum:
    """Sum of a bucket over ranks as a direct reduce-scatter + all-gather: every rank owns 1/world of the bucket,
    receives that slice from every peer in ONE all-to-all (on a fully connected xGMI node all peer links carry data at
    once, where a ring uses one link per direction), adds the world slices in rank order, and the reduced slices are
    all-gathered.  The rank-order sum makes the result identical on every rank and independent of the transport.
    Runs on a stream of its own; ``wait()`` makes the caller's stream wait for it."""

    _scratch = {}

    def __init__(self, part: torch.Tensor, group):
        world = dist.get_world_size(group)
        n = part.numel()
        if n % world:
            raise ValueError("direct all-reduce: the bucket length must be a multiple of the world size")
        key = (part.device, part.dtype, n, world)
        buf = _DirectSum._scratch.get(key)
        if buf is None:
            buf = _DirectSum._scratch[key] = torch.empty((world, n // world), dtype=part.dtype, device=part.device)
        self.stream = None
        if part.is_cuda:
            self.stream = _comm_stream(part.device)
            self.stream.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
            dist.all_to_all_single(buf.view(-1), part, group=group)         # row r: my slice as rank r holds it
            mine = part.view(world, -1)[dist.get_rank(group)]
            mine.copy_(buf[0])
            for r in range(1, world):                                        # rank order, whatever arrived first
                mine.add_(buf[r])
            dist.all_gather_into_tensor(part, mine.clone(), group=group)

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return True


_COMM_STREAMS = {}


def _comm_stream(device):
    s = _COMM_STREAMS.get(device)
    if s is None:
        s = _COMM_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def allreduce_buckets(flat: torch.Tensor, buckets, group=None, async_op: bool = False, algo: str = None):
    """Sum ``flat`` over ranks, one exchange per bucket (in place).  Returns the work handles when async.
    ``algo`` (default: environment MMSA_ALLREDUCE, else "rccl"): "rccl" = the backend's all-reduce (RCCL picks ring /
    tree / direct by topology); "direct" = :class:`_DirectSum`, an explicit reduce-scatter + all-gather built from one
    all-to-all and one all-gather per bucket."""
    algo = algo or os.environ.get("MMSA_ALLREDUCE", "rccl")
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"
    handles = []
    for lo, hi in buckets:
        if staged:
            h = _HostStagedSum(flat[lo:hi], group)
            if not async_op:
                h.wait()
        elif algo == "direct":
            # every rank owns 1/world of the bucket; what does not divide goes through the backend's all-reduce
            world = dist.get_world_size(group)
            cut = hi - (hi - lo) % world
            tail = None
            if cut < hi:
                tail = dist.all_reduce(flat[cut:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            h = _DirectSum(flat[lo:cut], group) if cut > lo else None
            if async_op:
                handles.extend(x for x in (h, tail) if x is not None)
                continue
            if h is not None:
                h.wait()
        else:
            h = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


class _HostStagedGather:
    """all-gather of a GPU tensor over a host-memory transport (see :class:`_HostStagedSum`)."""

    def __init__(self, out: torch.Tensor, src: torch.Tensor, group):
        self.out, self.group = out, group
        self.host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        self.host.copy_(src, non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()

    def wait(self):
        self.copied.synchronize()
        world = dist.get_world_size(self.group)
        parts = [torch.empty_like(self.host) for _ in range(world)]
        dist.all_gather(parts, self.host, group=self.group)
        self.out.copy_(torch.stack(parts).reshape(self.out.shape), non_blocking=True)
        return True


def allgather_rows(out: torch.Tensor, src: torch.Tensor, group=None, async_op: bool = False):
    """out[world, ...] <- every rank's src[...] (rank order).  Returns a handle with ``wait()`` when async."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        h = _HostStagedGather(out, src, group)
        if not async_op:
            h.wait()
        return h
    return dist.all_gather_into_tensor(out.view(-1, *src.shape[1:]) if src.dim() > 1 else out.view(-1), src,
                                       group=group, async_op=async_op)


def first_occurrence_mask(ids: torch.Tensor) -> torch.Tensor:
    """mask[i] = no j < i has ids[j] == ids[i] (static shapes: usable inside a captured launch sequence)."""
    s, idx = torch.sort(ids, stable=True)
    first = torch.ones_like(s, dtype=torch.bool)
    first[1:] = s[1:] != s[:-1]
    mask = torch.empty_like(first)
    mask[idx] = first
    return mask


class SparseRowExchange:
    """Sum over ranks of a row-sparse gradient table (the embedding's data gradient: a rank's batch touches at most
    B * L of the table's rows) without moving the table: every rank sends (token ids, the table rows at those ids) --
    ``n * E`` floats instead of ``vocab * E`` -- and adds the gathered rows of all ranks into a zeroed table, rank by
    rank in rank order.  A row that a rank's batch holds several times is sent once (first occurrence; the repeats
    travel as zero rows), ids outside the table travel as zero rows, so within one rank's contribution every table
    element receives at most one non-zero addend: the result is the same on every rank and equals the dense sum
    ``table_0 + table_1 + ...`` taken in rank order, bit for bit.
    Traffic per rank: sends n * E * 4 B (4 MiB for 32 x 128 tokens of width 256), receives (world - 1) times that;
    a ring all-reduce of the 31 MB table moves 2 (world - 1) / world * 31 MB each way."""

    def __init__(self, n: int, width: int, vocab: int, device, group=None):
        self.n, self.width, self.vocab, self.group = n, width, vocab, group
        self.world = dist.get_world_size(group)
        self.rows = torch.zeros((n, width), dtype=torch.float32, device=device)
        self.ids = torch.zeros((n,), dtype=torch.int64, device=device)
        self.rows_all = torch.zeros((self.world, n, width), dtype=torch.float32, device=device)
        self.ids_all = torch.zeros((self.world, n), dtype=torch.int64, device=device)

    def pack(self, table: torch.Tensor, ids: torch.Tensor):
        """Fills the send buffers from this rank's table (kernel launches on the current stream, static shapes)."""
        ids = ids.reshape(-1)
        if ids.numel() != self.n or table.shape != (self.vocab, self.width):
            raise ValueError("SparseRowExchange.pack: ids / table do not match the exchange's geometry")
        ok = (ids >= 0) & (ids < self.vocab)
        safe = torch.where(ok, ids, torch.zeros_like(ids))
        # first occurrences among the VALID ids only: an id outside the table must not take row 0's slot
        keep = ok & first_occurrence_mask(torch.where(ok, ids, torch.full_like(ids, -1)))
        self.ids.copy_(safe)
        torch.index_select(table, 0, safe, out=self.rows)
        self.rows.mul_(keep.unsqueeze(1).to(self.rows.dtype))

    def start(self):
        """Issues the two all-gathers (asynchronously); returns their handles."""
        return [allgather_rows(self.rows_all, self.rows, self.group, async_op=True),
                allgather_rows(self.ids_all, self.ids, self.group, async_op=True)]

    def finish(self, table: torch.Tensor, handles):
        """Waits for the gathers and rebuilds the table as the sum over ranks, in rank order."""
        for h in handles:
            h.wait()
        table.zero_()
        for r in range(self.world):
            table.index_add_(0, self.ids_all[r], self.rows_all[r])


def train_batches(n_items: int, per_rank_batch: int, epoch_seed: int, rank: int, world: int):
    """Index batches for one epoch: a seeded permutation cut into global batches of ``world * per_rank_batch``
    (incomplete tail dropped, as ``drop_last=True``); rank r takes positions ``r::world`` of each global batch."""
    g = torch.Generator().manual_seed(int(epoch_seed))
    perm = torch.randperm(n_items, generator=g)
    gb = world * per_rank_batch
    out = []
    for s in range(0, n_items - gb + 1, gb):
        out.append(perm[s:s + gb][rank::world].clone())
    return out


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None):
    """Copies rank ``src``'s module buffers (BatchNorm running statistics and batch counts) to every rank.

    Gradients are averaged over ranks, so parameters stay identical; BatchNorm's batch statistics are taken per rank
    (each rank normalises its own 32 utterances, as the single-process reference normalises its batch), so the
    RUNNING statistics drift apart between replicas.  Call this before evaluation or before writing a checkpoint on a
    rank other than ``src``, so that every replica scores with the statistics rank ``src`` would save."""
    rank, world = world_info(group)
    if world == 1:
        return
    direct = dist.get_backend(group) == "nccl"
    for b in module.buffers():
        if direct and b.is_cuda:
            dist.broadcast(b, group_src=src, group=group)
        else:
            # group_src keeps this independent of how the group's ranks map to global ranks
            h = b.detach().to("cpu", copy=True)
            dist.broadcast(h, group_src=src, group=group)
            if rank != src:
                b.copy_(h)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous inference shard [lo, hi): ceil(n/world) items per rank, the last ranks may get fewer or none."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def gather_predictions(probs: torch.Tensor, preds: torch.Tensor, n_items: int, group=None):
    """All-gather per-rank ``probs [n_r, C]`` / ``preds [n_r]`` (shards from :func:`shard_range`) back into dataset
    order on every rank.  Shards are padded to equal length for the collective and trimmed afterwards."""
    rank, world = world_info(group)
    if world == 1:
        return probs, preds
    per = (n_items + world - 1) // world
    C = probs.shape[1]
    pp = probs.new_zeros((per, C))
    pr = preds.new_zeros((per,))
    pp[:probs.shape[0]] = probs
    pr[:preds.shape[0]] = preds
    all_p = [torch.empty_like(pp) for _ in range(world)]
    all_r = [torch.empty_like(pr) for _ in range(world)]
    dist.all_gather(all_p, pp, group=group)
    dist.all_gather(all_r, pr, group=group)
    return torch.cat(all_p)[:n_items], torch.cat(all_r)[:n_items]
