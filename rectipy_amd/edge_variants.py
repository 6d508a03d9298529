"""Stateful / masked variants of the linear edge (reference rectipy/edges.py:68-174).  They carry per-step state
(delay buffers, recursive filters), so a network that uses one is evaluated step by step (every diff-eq node still
steps through the HIP kernels); they are k x n host-side projections, not part of the HBM-bound path.  Behaviour
follows the reference's ``forward`` methods, including the delay-buffer indexing of edges.py:91-94."""
from typing import Union

import numpy as np
import torch

from .edges import Linear


def _square(filter_weights, n_in, dtype):
    f = torch.as_tensor(np.asarray(filter_weights) if not isinstance(filter_weights, torch.Tensor) else filter_weights,
                        dtype=dtype)
    if f.dim() != 2 or f.shape[0] != n_in or f.shape[1] != n_in:
        raise ValueError("Intrinsic weights have to be a square matrix with the number of rows and columns matching"
                         "the number of inputs to the edge.")
    return f


class _Stateful(Linear):
    _extra = ()

    def to(self, device: str, **kwargs):
        super().to(device, **kwargs)
        for name in self._extra:
            setattr(self, name, getattr(self, name).to(device))
        return self


class LinearMasked(_Stateful):
    """``(weights * mask) @ x`` (reference edges.py:150-174)."""
    _extra = ("mask",)

    def __init__(self, n_in: int, n_out: int, mask: Union[np.ndarray, torch.Tensor], weights=None,
                 dtype: torch.dtype = torch.float64, detach: bool = True, **kwargs):
        mask = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask, dtype=dtype)
        if mask.shape[0] == n_in and mask.shape[1] == n_out:  # square masks are transposed too (edges.py:161-162)
            mask = mask.T
        elif mask.shape[0] != n_out or mask.shape[1] != n_in:
            raise ValueError("Shape of the provided mask does not match the input and output dimensions of the "
                             "source and target nodes.")
        self.mask = mask.contiguous()
        super().__init__(n_in, n_out, weights=weights, dtype=dtype, detach=detach)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        w = self.weights * self.mask
        return x @ w.T if x.dim() == 2 else w @ x


class LinearMemory(_Stateful):
    """Delay buffer in front of the projection (reference edges.py:68-94)."""
    _extra = ("buffer", "delays", "_cols", "_src")

    def __init__(self, n_in: int, n_out: int, delays: Union[np.ndarray, torch.Tensor], weights=None,
                 dtype: torch.dtype = torch.float64, detach: bool = True, **kwargs):
        delays = torch.as_tensor(np.asarray(delays) if not isinstance(delays, torch.Tensor) else delays).long()
        if len(delays) != n_in:
            raise ValueError("The number of delays must match the number of node inputs.")
        self.delays = delays
        self.buffer = torch.zeros((n_in, int(delays.max()) + 1), dtype=dtype)
        # `buffer[:, delays] = x` with repeated delays keeps the LAST matching input on the CPU and is undefined on a
        # GPU; resolve the duplicates once so that both behave like the CPU path of the reference
        cols = torch.unique(delays)
        self._cols = cols
        self._src = torch.stack([torch.nonzero(delays == c).max() for c in cols])
        super().__init__(n_in, n_out, weights=weights, dtype=dtype, detach=detach)

    def _shift(self):
        return self.buffer.roll(-1, 1)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() != 1:
            raise RuntimeError("delay edges are evaluated one step at a time")
        self.buffer = self._shift()
        self.buffer[:, self._cols] = x[self._src]  # == buffer[:, delays] = x, the reference's indexing (edges.py:93)
        return self.weights @ self.buffer[:, 0]


class LinearFilter(_Stateful):
    """Recursive filter y <- filter @ y + x in front of the projection (reference edges.py:97-120)."""
    _extra = ("filter", "y")

    def __init__(self, n_in: int, n_out: int, filter_weights, weights=None, dtype: torch.dtype = torch.float32,
                 detach: bool = True, **kwargs):
        self.filter = _square(filter_weights, n_in, dtype)
        self.y = torch.zeros(n_in, dtype=dtype)
        super().__init__(n_in, n_out, weights=weights, dtype=dtype, detach=detach)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() != 1:
            raise RuntimeError("filter edges are evaluated one step at a time")
        self.y = self.filter @ self.y + x
        return self.weights @ self.y


class LinearMemoryFilter(LinearMemory):
    """Delay buffer whose content is filtered every step (reference edges.py:123-147)."""
    _extra = ("buffer", "delays", "_cols", "_src", "filter")

    def __init__(self, n_in: int, n_out: int, delays, filter_weights, weights=None,
                 dtype: torch.dtype = torch.float64, detach: bool = True, **kwargs):
        self.filter = _square(filter_weights, n_in, dtype)
        super().__init__(n_in, n_out, delays=delays, weights=weights, dtype=dtype, detach=detach)

    def _shift(self):
        return self.filter @ self.buffer.roll(-1, 1)
