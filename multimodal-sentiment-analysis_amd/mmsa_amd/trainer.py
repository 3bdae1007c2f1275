"""Fused training step: front end -> forward -> backward -> (all-reduce) -> clip -> AdamW, no autograd.

Same step as the reference's loop body (`CNN implementation/multimodal_train.py:175-202`): zero grads, forward
with labels, backward, ``clip_grad_norm_(1.0)``, ``AdamW.step`` (lr / weight decay from `:481-485`), but

* gradients are written straight into one flat buffer (nothing to zero except the embedding rows),
* the L2 regulariser's gradient, the global-norm clip and AdamW are a few launches over flat buffers,
* no value is read back to the host inside the step: the loss, the clip coefficient, the step count and the two
  modality-dropout switches live in device memory, so the whole step can be captured in a HIP graph and
  replayed (``use_graph=True``) -- the reference synchronises three times per step (`:198-202`),
* with ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL), utterances are sharded
  across ranks and the flat gradient is summed over ranks; the L2 term is a function of the replicated parameters,
  so its gradient is added locally after the reduction and never communicated.  For modality 'both' the backward
  is split: head and text-branch gradients (the embedding table: 87 % of the bytes) are complete before the audio
  branch's backward starts, so their all-reduce runs on RCCL's stream underneath it; only the audio branch's
  5 MB are reduced on the critical path.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import audio as audio_engine, ops, parallel, text as text_engine
from .audio import fork
from .model import L2_LAMBDA, MultimodalSentimentModel
from .workspace import Grads, Workspace


# Only the capturing thread's calls are policed during capture: a process group's watchdog thread (or any other
# library thread) may touch the runtime meanwhile without invalidating the capture.
CAPTURE_MODE = "thread_local"


def dist_backend_is_stream_ordered(group) -> bool:
    """True when waiting for a collective's handle only orders GPU streams (RCCL); a host-memory transport's wait blocks
    the calling thread instead, so it is better called as late as possible."""
    import torch.distributed as dist
    return dist.get_backend(group) == "nccl"


class TrainStep:
    def __init__(self, model: MultimodalSentimentModel, lr: float = 1e-3, weight_decay: float = 1e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: Optional[float] = 1.0, frontend=None,
                 use_graph: bool = False, process_group=None, bucket_mb: float = 64.0, distributed: bool = True,
                 lr_from=None):
        """``lr_from``: a ``torch.optim`` optimiser whose first group's ``lr`` this step follows (read before every
        step).  The reference's schedulers (``ReduceLROnPlateau``, the cosine warm-up; multimodal_train.py:487-504,
        314-320) only accept a stock optimiser: keep the reference's ``optim.AdamW(model.parameters(), lr=...)`` as
        the object they drive -- its ``step()`` is never called -- and hand it over here."""
        self.model = model
        self.lr_from = lr_from
        if lr_from is not None:
            lr = float(lr_from.param_groups[0]["lr"])
        frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
        if frozen:
            # torch.optim.AdamW leaves a parameter without a gradient alone (no decay, no moment update); the fused step
            # updates the whole flat buffer.  Rather than train what the caller froze, refuse.
            raise ValueError(f"TrainStep updates every parameter; {len(frozen)} are frozen (requires_grad=False), e.g. "
                             f"{frozen[0]}: train a partly frozen model through loss.backward() and torch.optim.AdamW")
        self.arena = model._ensure_arena()
        dev = self.arena.device
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.g = self.arena.like()
        self.m = self.arena.like()
        self.v = self.arena.like()
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)
        self.grads = Grads({id(p): v for p, v in zip(self.arena.params, self.arena.views_of(self.g))})
        self.ws = Workspace(dev, persistent=True)
        self.frontend = frontend
        self._err_host = None
        if dev.type == "cuda":
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.ws.objs["text.err_host"] = self._err_host
        self.keep = torch.ones(2, dtype=torch.float32, device=dev)
        self._keep_table = torch.tensor([[1., 1.], [0., 1.], [1., 0.], [0., 0.]], device=dev)
        self.use_graph = use_graph
        self._graph = None
        self._baked = None
        self._norms_ready = False      # True once AdamW has left current per-chunk parameter sums behind
        self._pversion = None
        self._static = {}
        self._warm = 0
        self._nstep = 0
        self._replays = 0
        self.loss = None
        self.logits = None
        self.world = 1
        self.pg = process_group
        # the collective path is taken whenever a process group exists (also with a single rank, which keeps the
        # multi-rank code path testable on one GPU)
        # (distributed=False keeps a step local even inside an initialised job: e.g. instrumented steps on one rank)
        self.dist = distributed and torch.distributed.is_available() and torch.distributed.is_initialized()
        if self.dist:
            self.world = torch.distributed.get_world_size(process_group)
            import mmsa_amd
            if use_graph and not mmsa_amd.GRAPH_CHAIN_OK:
                # the multi-rank step replays a chain of graphs with the collectives between them, which is only
                # safe with the runtime's pre-recorded graph packets switched off (see mmsa_amd/__init__.py)
                import warnings
                warnings.warn("DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in effect when the GPU was initialised: the "
                              "multi-rank step launches its kernels eagerly instead of replaying a chain of graphs "
                              "(export the variable before starting the process to get graph replay)")
                self.use_graph = use_graph = False
            if self.world > 1 and model.modality == "both":
                # one modality-dropout decision per GLOBAL batch, as the reference makes one per batch
                # (multimodal_model.py:264-268): every rank draws from torch's host generator, so the generators must
                # be in the same state on all ranks (same seed, same construction sequence)
                parallel.assert_same_on_all_ranks(torch.get_rng_state(), "torch's host RNG state (seed every rank "
                                                  "identically before building the model)", process_group, dev)
        bucket = int(bucket_mb * 1024 * 1024 / 4)
        self._buckets = parallel.make_buckets(self.g.numel(), bucket)
        # overlapped exchange: [split, end) = text branch + head (ready early), [0, split) = audio branch
        # (costs ~0.1 ms of lost concurrency inside the step, so by default only when there is somebody to talk to;
        # MMSA_OVERLAP_ALLREDUCE=1 / 0 forces it on / off)
        want = os.environ.get("MMSA_OVERLAP_ALLREDUCE", "auto")
        self.overlap = self.dist and model.modality == "both" and (want == "1" or (want != "0" and self.world > 1))
        self._use_sparse = False
        self._sparse = None
        if self.overlap:
            n_audio = len(list(model.audio_model.parameters()))
            split = self.arena.offsets[n_audio]
            self._late_buckets = parallel.make_buckets(split, bucket)
            # The embedding table is the first tensor behind the audio branch and 81 % of all gradient bytes, but a
            # batch touches at most B * L of its 30 522 rows: exchanged as (ids, rows) its traffic is 4 MiB sent and
            # (world - 1) x 4 MiB received per rank instead of a 31 MB all-reduce.  Off unless MMSA_SPARSE_EMB=1: the
            # dense exchange already runs underneath the audio backward, while the sparse one is a chain of a dozen
            # small host-issued launches (sort, gather, two all-gathers, zero, ordered adds: 0.18 ms on its helper
            # stream) that cost 0.28 ms per step in a one-rank run (1.77 vs 1.49 ms); it pays only where the dense
            # exchange cannot be hidden (slow links, many ranks per link).
            self._use_sparse = os.environ.get("MMSA_SPARSE_EMB", "0") == "1"
            dense_from = self.arena.offsets[n_audio + 1] if self._use_sparse else split
            self._early_buckets = [(dense_from + lo, dense_from + hi) for lo, hi in
                                   parallel.make_buckets(self.g.numel() - dense_from, bucket)]

    # ------------------------------------------------------------------ one step's launch sequence
    def _forward_backward(self, audio, input_ids, labels, wave, phase: str = "all"):
        """Front end, forward, backward: leaves this rank's data gradient (already divided by the world size) in
        self.g.  Pure kernel launches on the current stream: capturable.  ``phase`` "early" stops after the head's
        and the text branch's backward, "late" runs the audio branch's backward of the same step."""
        model = self.model
        if phase == "late":
            model._backward(self._run, None, self.grads, self.ws, phase="late")
            return self._run.l2
        ops.stamp("step.start")
        if wave is not None:
            audio = self.ws.buf("step.spec", (wave.shape[0], 1, 64, 130))
            self.frontend(wave, out=audio)
            ops.stamp("fwd.frontend_done")
        pre_zero = phase == "early"
        if pre_zero:
            # the embedding-gradient table is cleared beside the forward pass instead of ahead of the text backward,
            # which sits on the critical path of the early exchange
            with fork(self.ws, self.arena.device, 1):
                self.grads.of(model.text_model.embedding.weight).zero_()
        run = model._run(audio, input_ids, labels, None, need_grad=True, ws=self.ws,
                         keep=(self.keep[0:1], self.keep[1:2]), norms_ready=self._norms_ready,
                         grad_scale=1.0 / self.world, step_state=self.state[0:1])
        if pre_zero:
            run.text.dtable_zeroed = True
        ops.stamp("step.forward_done")
        d = run.head.dlogits
        if self.world > 1 and not getattr(run.head, "grad_scaled", False):
            d.mul_(1.0 / self.world)            # sum over ranks of per-rank means -> global mean
        model._backward(run, d, self.grads, self.ws, phase=phase)
        self._run = run
        self.loss = run.loss
        self.logits = run.head.logits
        self.probs = run.head.probs
        return run.l2

    def _body(self, audio, input_ids, labels, wave):
        if self.overlap:
            l2 = self._forward_backward(audio, input_ids, labels, wave, phase="early")
            early = self._allreduce_early()
            self._forward_backward(audio, input_ids, labels, wave, phase="late")
            self._allreduce_late(early)
        else:
            l2 = self._forward_backward(audio, input_ids, labels, wave)
            if self.dist:
                self._allreduce()
        self._update(l2)

    def _allreduce(self):
        parallel.allreduce_buckets(self.g, self._buckets, self.pg)

    def _allreduce_early(self):
        """Text-branch and head gradients: issued on the collective's stream, not waited for yet.  With the row-sparse
        exchange the embedding table travels as (ids, rows); its gathered rows are summed into the table on a helper
        stream, beside the audio branch's backward."""
        handles = parallel.allreduce_buckets(self.g, self._early_buckets, self.pg, async_op=True)
        if self._use_sparse:
            emb = self.model.text_model.embedding.weight
            dtable = self.grads.of(emb)
            ids = self._cur_ids
            if self._sparse is None or self._sparse.n != ids.numel():
                self._sparse = parallel.SparseRowExchange(ids.numel(), emb.shape[1], emb.shape[0], self.arena.device,
                                                          self.pg)
            from .audio import FORK_ANY, side_stream
            self._sparse_pending = None
            if FORK_ANY and dtable.is_cuda and dist_backend_is_stream_ordered(self.pg):
                # the whole exchange (pack, two all-gathers, zero + ordered adds) on a helper stream behind the early
                # backward: nothing of it sits between the early and the late part of the step
                helper = side_stream(self.arena.device, 3)
                helper.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(helper):
                    self._sparse.pack(dtable, ids)
                    self._sparse.finish(dtable, self._sparse.start())
                self._sparse_pending = helper
            else:
                self._sparse.pack(dtable, ids)
                self._sparse_pending = (dtable, self._sparse.start())
        return handles

    def _allreduce_late(self, early):
        """Audio-branch gradients, then make the compute stream wait for both exchanges."""
        late = parallel.allreduce_buckets(self.g, self._late_buckets, self.pg, async_op=True)
        for h in list(early) + list(late):
            h.wait()
        if self._use_sparse:
            if isinstance(self._sparse_pending, tuple):
                self._sparse.finish(*self._sparse_pending)
            else:
                torch.cuda.current_stream().wait_stream(self._sparse_pending)

    def _update(self, l2: bool):
        """Clip + AdamW over the flat buffers.  self.g holds the DATA gradient only: the regulariser's gradient
        0.001 * p / ||p|| (per-tensor norms were computed by the forward pass) is folded into both kernels."""
        a = self.arena
        do_clip = self.max_norm is not None
        tn, lam = (a.tnorm, L2_LAMBDA) if l2 else (None, 0.0)
        ops.stamp("step.backward_done")
        if do_clip:      # the norm kernel's second launch also advances the step count and derives the clip coefficient
            ops.grad_norm_finish(self.g, a.table, self.state, self.betas[0], self.betas[1], self.max_norm, True,
                                 a.flat if l2 else None, tn, lam, sparse=self._sparse_rows())
        # AdamW also leaves the per-chunk sums of squares of the UPDATED parameters behind: the next step's
        # regulariser norms need no pass of their own over the 38 MB parameter buffer
        ops.adamw_step(a.flat, self.g, self.m, self.v, a.table, self.state, self.lr, self.betas[0], self.betas[1],
                       self.eps, self.wd, self.max_norm if do_clip else 0.0, do_clip, tn, lam, emit_param_sums=l2,
                       scalars_ready=do_clip)
        self._norms_ready = l2
        ops.stamp("step.end")

    def _sparse_rows(self):
        """The embedding table's gradient holds this batch's token rows and zeros elsewhere: while it is this rank's
        own (no exchange has added other ranks' rows) the clip norm reads those rows only."""
        tm = self.model.text_model
        st = getattr(self._run, "text", None) if getattr(self, "_run", None) is not None else None
        if tm is None or st is None or self.dist or os.environ.get("MMSA_SPARSE_NORM", "1") == "0":
            return None
        emb = tm.embedding.weight
        t = next(i for i, p in enumerate(self.arena.params) if p is emb)
        n = st.ids.numel()
        return t, emb.shape[1], st.ids, st.groups[2 * n:3 * n]

    @staticmethod
    def _copy_inputs(pending):
        for i in range(0, len(pending), 4):
            ops.copy_multi(pending[i:i + 4])

    # ------------------------------------------------------------------ public API
    def step(self, audio=None, input_ids=None, labels=None, wave=None):
        """Runs one optimisation step.  Returns the device tensor holding the loss (no host sync)."""
        # kernels are launched on the current device: the model's, for the duration of the step
        with torch.cuda.device(self.arena.device):
            return self._step(audio, input_ids, labels, wave)

    def _step(self, audio, input_ids, labels, wave):
        model = self.model
        if labels is None:
            raise ValueError("labels are required for a training step")
        if wave is not None and self.frontend is None:
            raise ValueError("pass frontend=LogMelFrontEnd(...) to train from waveforms")
        if not model.training:
            raise RuntimeError("TrainStep.step() needs model.train()")
        if self.lr_from is not None:
            self.set_lr(self.lr_from.param_groups[0]["lr"])
        if self._graph is not None and self._baked != self._scalars():
            self._graph = None      # an optimiser scalar was assigned to since the capture, which holds the old value
        self._check_storage(full=self._graph is None or self._nstep % self.CHECK_EVERY == 0)
        # parameter sums left by the previous step stay valid only if nobody else wrote the parameters since
        # (load_state_dict, a user optimiser, ...): torch's version counters tell
        pv = sum(p._version for p in self.arena.params)
        if pv != self._pversion:
            self._pversion = pv
            if self._graph is None:
                self._norms_ready = False
            elif self._norms_ready and model.modality == "both":
                # the captured sequence trusts the stored sums: refresh them once, outside the graph
                ops.param_norms(self.arena.flat, self.arena.table, self.arena.tnorm, None, False)
        pending = []          # (dst, src) device copies of this step's inputs: one launch for all of them
        if model.modality == "both":
            da, dt = model.fusion_layer.draw_modality_dropout()     # host RNG, audio first, as the reference
            pending.append((self.keep, self._keep_table[int(da) + 2 * int(dt)]))
        dev = self.arena.device
        if wave is not None:
            audio = None
        ins = {"audio": audio, "input_ids": input_ids, "labels": labels, "wave": wave}
        if model.modality == "text":
            ins["audio"] = ins["wave"] = None
        if model.modality == "audio":
            ins["input_ids"] = None
        model._check_inputs(ins["audio"] if ins["wave"] is None else ins["wave"], ins["input_ids"], labels,
                            audio_dims=(3, 4) if ins["wave"] is None else (2,))
        # a token id outside the embedding table raises IndexError in the reference.  Ids that arrive in host memory
        # are checked here; device-resident ids are checked by the kernels (they skip such tokens and raise a sticky
        # flag).  Every step copies that flag to pinned host memory, so the host sees it without waiting for the GPU:
        # the IndexError comes at the start of the first step() that finds the copy of an offending step landed
        # (normally the next one or the one after), and from check_ids() / state_dict(), which synchronise
        # (class labels outside [0, C) likewise: nn.CrossEntropyLoss raises IndexError in the reference)
        self._nstep += 1
        if ins["input_ids"] is not None and not ins["input_ids"].is_cuda:
            text_engine.check_ids_host(ins["input_ids"], model.text_model.embedding.num_embeddings)
        if not labels.is_cuda:
            text_engine.check_labels_host(labels, model.classifier[len(model.classifier) - 1].weight.shape[0])
        if (self._err_host is not None and int(self._err_host[0])) or self._nstep % self.CHECK_EVERY == 0:
            self.check_ids()
        for k, t in ins.items():
            if t is not None:
                want = torch.int64 if k in ("input_ids", "labels") else torch.float32
                ins[k] = t.to(device=dev, dtype=want, non_blocking=True).contiguous()
        if not self.use_graph:
            self._copy_inputs(pending)
            self._cur_ids = ins["input_ids"]
            self._body(ins["audio"], ins["input_ids"], ins["labels"], ins["wave"])
            return self.loss
        # graph mode: inputs are copied into static buffers, the captured launch sequences are replayed.
        # One graph holds the whole step on a single GPU; with several ranks the step is two graphs
        # (forward+backward, clip+AdamW) with the RCCL all-reduce issued eagerly between them, or three for
        # modality 'both' (forward + early backward | audio backward | clip+AdamW) with the first exchange
        # running underneath the second graph.
        if self._graph is not None and any(t is not None and (self._static.get(k) is None
                                                              or self._static[k].shape != t.shape)
                                           for k, t in ins.items()):
            # a batch of another shape than the captured one (the reference's training loader drops the last, short
            # batch, multimodal_train.py:441; a caller's own loader may not): this step is launched kernel by kernel,
            # the captured sequence stays as it is for the batches of the usual shape
            if self.dist:
                raise ValueError("graph mode on several ranks needs fixed input shapes (drop_last)")
            self._copy_inputs(pending)
            self._cur_ids = ins.get("input_ids")
            self._body(ins["audio"], ins["input_ids"], ins["labels"], ins["wave"])
            return self.loss
        for k, t in ins.items():
            if t is None:
                continue
            s = self._static.get(k)
            if s is None or s.shape != t.shape:
                self._static[k] = t.clone()
            elif s.data_ptr() != t.data_ptr():
                pending.append((s, t))
        self._copy_inputs(pending)
        st = self._static
        self._cur_ids = st.get("input_ids")
        args = (st.get("audio"), st.get("input_ids"), st.get("labels"), st.get("wave"))
        if self._graph is None:
            if self._warm < 2:                 # eager warm-up: sizes every workspace buffer before capture
                self._warm += 1
                self._body(*args)
                return self.loss
            torch.cuda.synchronize()
            self._baked = self._scalars()
            g = torch.cuda.CUDAGraph()
            if not self.dist:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    self._body(*args)
                self._graph = (g, None, None)
            elif self.overlap:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    l2 = self._forward_backward(*args, phase="early")
                gl = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gl, pool=g.pool(), capture_error_mode=CAPTURE_MODE):
                    self._forward_backward(*args, phase="late")
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=CAPTURE_MODE):
                    self._update(l2)
                self._graph = (g, gl, g2)
            else:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    l2 = self._forward_backward(*args)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=CAPTURE_MODE):
                    self._update(l2)
                self._graph = (g, None, g2)
            # capture only records: the replays below execute this step
        g, gl, g2 = self._graph
        g.replay()
        if gl is not None:
            early = self._allreduce_early()
            gl.replay()
            self._allreduce_late(early)
            g2.replay()
        elif g2 is not None:
            self._allreduce()
            g2.replay()
        if g2 is not None and self._replays < 3:
            # a chain of graphs: make sure the first replays produced numbers (costs three synchronisations in total)
            self._replays += 1
            vals = self.state[4:5].isfinite() & self.loss.isfinite().all()
            if not bool(vals.item()):
                raise RuntimeError("replaying the multi-rank step as a chain of HIP graphs produced a non-finite loss or "
                                   "gradient norm; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before "
                                   "the process first touches the GPU")
        return self.loss

    CHECK_EVERY = 256

    def _scalars(self):
        """What a captured step holds by value: the optimiser's scalars, the dropout rates, and whether the audio
        branch's BatchNorm layers use batch statistics (``model.audio_model.eval()`` under a training model freezes
        them)."""
        m = self.model
        bn = None if m.audio_model is None else audio_engine.bn_training(m.audio_model)
        if m.modality == "audio":
            bn = (bn, m.classifier[2].training, m.classifier[6].training)
        return (float(self.lr), float(self.wd), float(self.eps), tuple(float(b) for b in self.betas),
                None if self.max_norm is None else float(self.max_norm), tuple(m._dropouts), bn)

    def _check_storage(self, full: bool):
        """The step updates the flat buffer the model's parameters lived in when this object was built.  After
        ``model.to(...)`` / ``.float()`` (every parameter re-allocated) or an exchanged layer (other Parameter objects)
        they live elsewhere, and the update would move numbers nobody reads: refuse.  Cheap form on every step (the
        model still points at this storage, the first tensor still sits at its start), the walk over all tensors before
        a capture, on eagerly launched steps and every CHECK_EVERY replayed ones."""
        a = self.arena
        ok = self.model.__dict__.get("_arena") is a and a.params[0].data_ptr() == a.flat.data_ptr()
        if ok and full:
            cur = list(self.model.parameters())
            ok = len(cur) == len(a.params) and all(x is y for x, y in zip(cur, a.params)) and a.owns()
        if not ok:
            raise RuntimeError("the model's parameters no longer live in the storage this TrainStep updates (the model "
                               "was moved or converted, or a layer was exchanged, after the TrainStep was built): "
                               "build a new TrainStep, carrying the optimiser state over with state_dict() / "
                               "load_state_dict()")

    def check_ids(self):
        """Raises IndexError if any step since the last call saw a token id outside the embedding table (those
        tokens contributed zero rows and no gradient) or a class label outside [0, C) (those rows contributed no loss
        term).  Synchronises."""
        if self._err_host is not None:
            torch.cuda.synchronize(self.arena.device)
            self._err_host.zero_()
        text_engine.check_ids(self.ws.buf("text.err", 1, torch.int32))

    @torch.no_grad()
    def grad_norm(self) -> torch.Tensor:
        return self.state[4]

    def state_dict(self):
        """Optimiser state in ``torch.optim.AdamW.state_dict()`` layout (multimodal_train.py:347 saves that).  Raises
        IndexError instead if a step since the last check saw a token id outside the embedding table (the reference
        would have raised before that update: nothing of it should reach a checkpoint)."""
        self.check_ids()
        step = float(self.state[0].item())
        ms, vs = self.arena.views_of(self.m), self.arena.views_of(self.v)
        state = {i: {"step": torch.tensor(step), "exp_avg": ms[i].clone(), "exp_avg_sq": vs[i].clone()}
                 for i in range(len(ms))} if step > 0 else {}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "params": list(range(len(ms)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Inverse of :meth:`state_dict`: accepts what ``torch.optim.AdamW.state_dict()`` of the reference's optimiser
        (same parameter order, one group) or this class wrote."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.arena.params):
            raise ValueError("optimizer state does not match this model: expected one group of "
                             f"{len(self.arena.params)} parameters")
        grp = groups[0]
        self.wd, self.eps, self.betas = grp["weight_decay"], grp["eps"], tuple(grp["betas"])
        self.set_lr(grp["lr"])
        ms, vs = self.arena.views_of(self.m), self.arena.views_of(self.v)
        state = sd["state"]
        step = 0.0
        self.m.zero_()
        self.v.zero_()
        for i, key in enumerate(grp["params"]):
            st = state.get(key)
            if st is None:
                continue
            ms[i].copy_(st["exp_avg"].reshape(ms[i].shape))
            vs[i].copy_(st["exp_avg_sq"].reshape(vs[i].shape))
            step = max(step, float(st["step"]))
        self.state[0] = step

    def set_lr(self, lr: float):
        """Learning-rate change from a scheduler (multimodal_train.py:314-320).  Scalars are baked into a captured
        launch sequence, so graph mode re-captures on the next step."""
        if float(lr) != self.lr or self._graph is None:
            self.lr = float(lr)
            self._graph = None

