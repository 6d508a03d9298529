"""Stateful / masked variants of the linear edge (reference rectipy/edges.py:68-174).  They carry per-step state
(delay buffers, recursive filters); they are k x n host-side projections, not part of the HBM-bound path.  Behaviour
follows the reference's ``forward`` methods, including the delay-buffer indexing of edges.py:91-94.

On an ACYCLIC edge such a map is a causal function of the source's window: ``forward_window(X[T, n_in])`` evaluates it
for all T steps of a window (state carried across windows exactly as the per-step ``forward`` carries it), so the
populations on either side still integrate as ONE fused window each (Network._edge_window).  The edge's MEMORY -- delay
buffer, recursive filter -- is a device stage of the library (rp_edge_mem / rp_edge_window, rectipy_amd/csrc/rnn_edges.hip):
one call per window on an acyclic edge, advanced once per Euler step inside rp_forward_coupled when the edge lies in a
cycle (FeedbackNetwork).  The per-step ``forward`` methods below restate the reference's classes in torch and stay the
path a gradient through the memory takes."""
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
    _backend = None   # test seam (Network(backend=...)): the library backend of the edge's memory stage on CPU tensors
    _stage_kind = None  # "delay" | "filter" | "delay_filter": the rp_edge_mem kind of the edge's memory (engine.EdgeMemory)

    def to(self, device: str, **kwargs):
        super().to(device, **kwargs)
        for name in self._extra:
            setattr(self, name, getattr(self, name).to(device))
        return self

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        """One step of the edge's memory: takes the source's output x, returns the vector the projection is applied to."""
        raise NotImplementedError

    # ---- the memory as a device stage (rp_edge_mem: include/rectipy_hip.h, rectipy_amd/csrc/rnn_edges.hip) ------------------
    def _stage_state(self) -> torch.Tensor:
        """The edge's memory in the stage's layout (engine.EdgeMemory)."""
        raise NotImplementedError

    def _stage_restore(self, state: torch.Tensor):
        raise NotImplementedError

    def _stage_args(self) -> dict:
        return {}

    def memory_stage(self, no_grad_input: bool = True):
        """The edge's memory as an :class:`rectipy_amd.engine.EdgeMemory` loaded with the edge's current state, or None when
        the library cannot take it: a gradient would have to pass through the memory, tensors that are neither float32 on
        a GPU nor (tests) served by an injected backend, a delay buffer whose rows differ.  After the library has advanced
        the stage, :meth:`stage_done` writes the state back into the edge's own tensors (`buffer`, `y`), so per-step
        `forward` calls and library windows can alternate."""
        if self._stage_kind is None or not no_grad_input:
            return None
        W = self.weights
        if W.dtype != torch.float32 or (W.device.type != "cuda" and self._backend is None):
            return None
        state = self._stage_state()
        if state is None:
            return None
        from .engine import EdgeMemory
        mem = getattr(self, "_mem", None)
        if mem is None or mem.device != W.device:
            mem = EdgeMemory(self._stage_kind, self.n_in, W.device, backend=self._backend, **self._stage_args())
            self._mem = mem
        F = getattr(self, "filter", None)
        if F is not None and (mem.F is None or mem.F.data_ptr() != F.data_ptr()):
            mem.set_filter(F)
        mem.load(state)
        return mem

    def stage_done(self, mem):
        self._stage_restore(mem.current)

    def forward_window(self, X: torch.Tensor) -> torch.Tensor:
        """The edge's outputs for the T consecutive inputs X[t] ([T, n_in] -> [T, n_out]); the edge's state afterwards
        is what T calls of ``forward`` leave behind.

        The T steps of the edge's MEMORY (delay buffer, recursive filter) are one library call (rp_edge_window): narrow edges
        -- an input node's few channels -- one single-workgroup launch for the whole window, wide ones -- an edge between two
        populations, whose filter matrix is as large as a recurrent one -- a streaming pass over the filter per step, issued
        from C.  The projection of the window is ONE product on the device (a gradient with respect to the weights passes
        through it).  Only when a gradient has to pass through the memory itself (an input that requires grad), or without
        a GPU, the T calls of ``forward`` are made one by one."""
        T = X.shape[0]
        if T == 0:
            return X.new_zeros((0, self.weights.shape[0]))
        mem = self.memory_stage(no_grad_input=not (X.requires_grad and torch.is_grad_enabled()))
        if mem is None:
            return torch.stack([self.forward(X[t]) for t in range(T)])
        U = mem.window(X.detach())
        self.stage_done(mem)
        return U @ self.weights.T


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

    _stage_kind = "delay"

    def _col_chan(self) -> np.ndarray:
        """Per buffer column the input channel `buffer[:, delays] = x` leaves there (-1: none): rp_edge_mem.col_chan."""
        cc = -np.ones(self.buffer.shape[1], dtype=np.int32)
        cc[self._cols.cpu().numpy()] = self._src.cpu().numpy()
        return cc

    def _stage_args(self) -> dict:
        return dict(depth=self.buffer.shape[1], col_chan=self._col_chan())

    def _stage_state(self):
        # the stage keeps ONE row of the buffer: every write goes to all rows alike, so the rows are identical unless
        # someone set them apart
        if not bool((self.buffer == self.buffer[:1]).all()):
            return None
        return self.buffer[0]

    def _stage_restore(self, state):
        self.buffer = state.to(self.buffer.dtype).reshape(1, -1).expand(self.buffer.shape[0], -1).contiguous()

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

    _stage_kind = "filter"

    def _stage_args(self) -> dict:
        return dict(F=self.filter)

    def _stage_state(self):
        return self.y

    def _stage_restore(self, state):
        self.y = state.to(self.y.dtype).clone()

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

    _stage_kind = "delay_filter"

    def _stage_args(self) -> dict:
        return dict(F=self.filter, depth=self.buffer.shape[1], col_chan=self._col_chan())

    def _stage_state(self):
        return self.buffer.T   # row c of the stage = column c of the buffer

    def _stage_restore(self, state):
        self.buffer = state.to(self.buffer.dtype).T.contiguous()

    def _advance(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's step, as its classes execute it (edges.py:143-146): the filtered buffer is shifted and x is
        written at the delays -- and then ``super().forward(self.buffer[:, 0])`` is ``LinearMemory.forward``, which shifts
        the buffer AGAIN and writes that column-0 vector at the delays before projecting column 0.  With a zero delay the
        second phase returns the same column 0 (fixture G27 only has such delays); without one it does not -- found by
        scripts/fuzz_edges_vs_reference.py against the reference's own class."""
        self.buffer = self.filter @ self.buffer.roll(-1, 1)
        self.buffer[:, self._cols] = x[self._src]
        return super()._advance(self.buffer[:, 0].clone())
