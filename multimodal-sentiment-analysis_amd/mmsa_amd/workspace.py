"""Named device buffers reused across steps (static addresses: graph-capture friendly, no allocator traffic)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class Workspace:
    """``persistent=True`` caches each (name, shape, dtype) buffer; ``False`` hands out fresh tensors
    (used when autograd may keep several forward passes alive at once)."""

    def __init__(self, device, persistent: bool = True):
        self.device = torch.device(device)
        self.persistent = persistent
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self.objs: Dict[str, object] = {}      # small host-side plans (descriptor tables) that live with the buffers

    def buf(self, name: str, shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        if not self.persistent:
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        key = (name, shape, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        elif zero:
            t.zero_()
        return t

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._bufs.values())


class Grads:
    """Where each parameter's gradient is written.  Pre-filled with views of a flat gradient buffer by the
    trainer; otherwise tensors are allocated on demand (autograd path)."""

    def __init__(self, mapping=None):
        self.map = dict(mapping) if mapping else {}

    def of(self, p: torch.Tensor, zero: bool = False) -> torch.Tensor:
        g = self.map.get(id(p))
        if g is None:
            g = (torch.zeros_like if zero else torch.empty_like)(p.data)
            self.map[id(p)] = g
        return g

    def get(self, p):
        return self.map.get(id(p))
