"""Stateful / masked variants of the linear edge (reference rectipy/edges.py:68-174).  They carry per-step state
(delay buffers, recursive filters); they are k x n host-side projections, not part of the HBM-bound path.  Behaviour
follows the reference's ``forward`` methods, including the delay-buffer indexing of edges.py:91-94.

On an ACYCLIC edge such a map is a causal function of the source's window: ``forward_window(X[T, n_in])`` evaluates it
for all T steps of a window (state carried across windows exactly as the per-step ``forward`` carries it), so the
populations on either side still integrate as ONE fused window each (Network._edge_window).  The delay buffer has a
closed form over the window; the recursive filters are T small matrix-vector products in step order -- the reference's
own arithmetic, decoupled from the populations' kernel launches."""
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

    HOST_RECURRENCE_MAX_N_IN = 64  # narrow edges (an input node's few channels): the T-step recurrence runs on the host
    _state = ()  # the tensors `_advance` reads and replaces (besides its constants in `_extra`)

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        """One step of the edge's memory: takes the source's output x, returns the vector the projection is applied to."""
        raise NotImplementedError

    def forward_window(self, X: torch.Tensor) -> torch.Tensor:
        """The edge's outputs for the T consecutive inputs X[t] ([T, n_in] -> [T, n_out]); the edge's state afterwards
        is what T calls of ``forward`` leave behind.  Default: exactly those T calls.

        On a GPU every call is a handful of tiny launches (~10 us each, whatever the size).  For a NARROW edge without a
        gradient the T steps of the edge's MEMORY (delay buffer, recursive filter: a few KB) therefore run on host copies
        of its tensors (microseconds per step), and the projection of the whole window is ONE product on the device.
        Same formulas in float32; equal to the device loop up to the rounding order of the small matrix products."""
        T = X.shape[0]
        if T == 0:
            return X.new_zeros((0, self.weights.shape[0]))
        narrow = X.is_cuda and X.shape[1] <= self.HOST_RECURRENCE_MAX_N_IN and T > 1 and not X.requires_grad and \
            not self.weights.requires_grad and not any(getattr(self, n).requires_grad for n in self._extra
                                                       if isinstance(getattr(self, n), torch.Tensor))
        if not narrow:
            return torch.stack([self.forward(X[t]) for t in range(T)])
        dev = X.device
        names = tuple(self._extra)
        on_dev = {n: getattr(self, n) for n in names}
        host = {n: t.detach().cpu() for n, t in on_dev.items()}
        try:
            for n, t in host.items():
                setattr(self, n, t)
            Xh = X.detach().cpu()
            U = torch.stack([self._advance(Xh[t]) for t in range(T)])
        finally:
            for n in names:  # state the steps REPLACED (buffer, y) goes back to the device; everything else is the original object
                cur = getattr(self, n)
                setattr(self, n, on_dev[n] if cur is host[n] else cur.to(dev))
        return U.to(device=dev, dtype=self.weights.dtype) @ self.weights.T


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

    def forward_window(self, X: torch.Tensor) -> torch.Tensor:
        return self.forward(X)  # stateless: one product for the whole window


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

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        self.buffer = self._shift()
        self.buffer[:, self._cols] = x[self._src]  # == buffer[:, delays] = x, the reference's indexing (edges.py:93)
        return self.buffer[:, 0]

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() != 1:
            raise RuntimeError("delay edges are evaluated one step at a time")
        return self.weights @ self._advance(x)

    def forward_window(self, X: torch.Tensor) -> torch.Tensor:
        """Closed form of T steps of the plain delay buffer.  ``buffer[:, delays] = x`` writes the SAME row into every
        buffer row (edges.py:93), so one row b[0..D] describes the buffer: a step shifts it left (b[c] <- b[c+1]) and
        sets b[d] = x[src(d)] for every distinct delay d.  Hence column c at the end of step t holds
        x_{t-(d-c)}[src(d)] with d the smallest delay >= c, or the initial b[c+t+1] while t < d - c; the output is
        column 0.  Pure selection -- no arithmetic besides the projection, which runs once for the window."""
        if type(self) is not LinearMemory or X.shape[0] == 0 or X.requires_grad or \
                not bool((self.buffer == self.buffer[:1]).all()):
            return super().forward_window(X)  # filtered buffers, rows that differ, autograd: the T-step loop
        T, D1 = X.shape[0], self.buffer.shape[1]
        dev = self.buffer.device
        cols = self._cols.to(dev)
        c = torch.arange(D1, device=dev)
        nxt = torch.searchsorted(cols, c)                  # index of the smallest delay >= c (the largest delay is D)
        d, src = cols[nxt], self._src.to(dev)[nxt]
        lag = d - c                                        # steps a value needs from its write column d to column c
        b0 = self.buffer[0]

        def column(cc, t):  # value of column cc (index tensor) at the end of step t (tensor of the same shape)
            back = t - lag[cc]
            from_x = back >= 0
            xv = X[back.clamp(min=0), src[cc]]
            init = b0[(cc + t + 1).clamp(max=D1 - 1)]
            return torch.where(from_x, xv, init.to(X.dtype))

        t_all = torch.arange(T, device=dev)
        out0 = column(torch.zeros(T, dtype=torch.long, device=dev), t_all)          # buffer[:, 0] after every step
        last = column(c, torch.full((D1,), T - 1, dtype=torch.long, device=dev))    # the whole buffer after the last step
        self.buffer = last.to(self.buffer.dtype).reshape(1, D1).expand(self.buffer.shape[0], D1).contiguous()
        # every row of the buffer is the same, so buffer[:, 0] is out0[t] repeated n_in times
        return out0.reshape(T, 1).expand(T, self.weights.shape[1]).to(self.weights.dtype) @ self.weights.T


class LinearFilter(_Stateful):
    """Recursive filter y <- filter @ y + x in front of the projection (reference edges.py:97-120)."""
    _extra = ("filter", "y")

    def __init__(self, n_in: int, n_out: int, filter_weights, weights=None, dtype: torch.dtype = torch.float32,
                 detach: bool = True, **kwargs):
        self.filter = _square(filter_weights, n_in, dtype)
        self.y = torch.zeros(n_in, dtype=dtype)
        super().__init__(n_in, n_out, weights=weights, dtype=dtype, detach=detach)

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        self.y = self.filter @ self.y + x
        return self.y

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() != 1:
            raise RuntimeError("filter edges are evaluated one step at a time")
        return self.weights @ self._advance(x)


class LinearMemoryFilter(LinearMemory):
    """Delay buffer whose content is filtered every step (reference edges.py:123-147)."""
    _extra = ("buffer", "delays", "_cols", "_src", "filter")

    def __init__(self, n_in: int, n_out: int, delays, filter_weights, weights=None,
                 dtype: torch.dtype = torch.float64, detach: bool = True, **kwargs):
        self.filter = _square(filter_weights, n_in, dtype)
        super().__init__(n_in, n_out, delays=delays, weights=weights, dtype=dtype, detach=detach)

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's step, as its classes execute it (edges.py:143-146): the filtered buffer is shifted and x is
        written at the delays -- and then ``super().forward(self.buffer[:, 0])`` is ``LinearMemory.forward``, which shifts
        the buffer AGAIN and writes that column-0 vector at the delays before projecting column 0.  With a zero delay the
        second phase returns the same column 0 (fixture G27 only has such delays); without one it does not -- found by
        scripts/fuzz_edges_vs_reference.py against the reference's own class."""
        self.buffer = self.filter @ self.buffer.roll(-1, 1)
        self.buffer[:, self._cols] = x[self._src]
        return super()._advance(self.buffer[:, 0].clone())
