"""Flat parameter storage.

All trainable tensors of a model live back to back in one fp32 buffer (each start aligned to 256 floats,
gaps stay zero), so the regulariser, the gradient-norm clip, the optimiser and the data-parallel
all-reduce are each a handful of launches over one buffer instead of one per tensor (the reference loops
over 70 tensors three times per step: `CNN implementation/multimodal_model.py:460-461`,
`multimodal_train.py:194-195`).  ``nn.Parameter`` objects keep their reference shapes and names: their
``.data`` is re-pointed at a view of the flat buffer, so ``state_dict`` / ``load_state_dict`` are unchanged.
"""
from __future__ import annotations

from typing import List

import torch

ALIGN = 256          # floats
CHUNK = 16384        # floats per reduction chunk


class ChunkTable:
    """Maps fixed-size chunks of the flat buffer to tensors (device arrays consumed by optim.hip)."""

    def __init__(self, offsets: List[int], sizes: List[int], device):
        off, ln, tid, first = [], [], [], []
        for t, (o, n) in enumerate(zip(offsets, sizes)):
            first.append(len(off))
            pos = 0
            while pos < n:
                c = min(CHUNK, n - pos)
                off.append(o + pos)
                ln.append(c)
                tid.append(t)
                pos += c
        first.append(len(off))
        self.nchunks = len(off)
        self.ntensors = len(sizes)
        self.first_host = first          # tensor t owns chunks first_host[t] .. first_host[t + 1]
        self.chunk_off = torch.tensor(off, dtype=torch.int64, device=device)
        self.chunk_len = torch.tensor(ln, dtype=torch.int32, device=device)
        self.chunk_tensor = torch.tensor(tid, dtype=torch.int32, device=device)
        self.tensor_first_chunk = torch.tensor(first, dtype=torch.int32, device=device)
        self.chunk_ss = torch.zeros(self.nchunks, dtype=torch.float64, device=device)      # gradient sums
        self.chunk_ss_p = torch.zeros(self.nchunks, dtype=torch.float64, device=device)    # parameter sums


class ParamArena:
    def __init__(self, params: List[torch.nn.Parameter], device):
        self.params = list(params)
        self.offsets, self.sizes = [], []
        pos = 0
        for p in self.params:
            self.offsets.append(pos)
            self.sizes.append(p.numel())
            pos += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = max(pos, ALIGN)
        self.device = torch.device(device)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.views = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.views.append(v)
        self.table = ChunkTable(self.offsets, self.sizes, self.device)
        self.tnorm = torch.zeros(len(self.params), dtype=torch.float32, device=self.device)

    def owns(self) -> bool:
        """True while every parameter still aliases its slot (``module.to()`` / ``.float()`` re-allocate)."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o and p.device == self.flat.device
                   for p, o in zip(self.params, self.offsets))

    def like(self) -> torch.Tensor:
        return torch.zeros_like(self.flat)

    def views_of(self, flat: torch.Tensor):
        return [flat[o:o + n].view(p.shape) for p, o, n in zip(self.params, self.offsets, self.sizes)]
