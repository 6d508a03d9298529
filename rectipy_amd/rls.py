"""Recursive-least-squares readout edge (reference ``RLS``, rectipy/edges.py:177-234) on the GPU.

``update(x, y, y_hat)`` keeps the reference's per-step signature; ``sweep(X, Y, stride)`` trains over a whole
window of stored edge inputs in one library call (``rp_rls_sweep``), fusing the rank-1 update of the
state-error correlation matrix ``P`` with the mat-vec of the next update step.
"""
from typing import Optional, Union

import numpy as np
import torch

from . import _abi
from .edges import Linear


class RLS(Linear):

    _tensors = ["weights", "P"]

    def __init__(self, n_in: int, n_out: int, weights: Union[np.ndarray, torch.Tensor] = None,
                 dtype: torch.dtype = torch.float32, beta: float = 1.0, alpha: float = 1.0, backend=None, **kwargs):
        if beta > 1 or beta < 0:
            raise ValueError("Parameter beta should be a positive scalar between 0 and 1.")
        if alpha < 0:
            raise ValueError("Parameter alpha should be a positive scalar.")
        if weights is None:
            weights = torch.zeros((n_out, n_in), dtype=torch.float32)
        super().__init__(n_in, n_out, weights=weights, dtype=torch.float32, detach=True)
        self.train_params = []
        self.beta = beta ** (-1)  # the reference stores the inverse forgetting rate (edges.py:220)
        self._P = alpha * torch.eye(n_in, dtype=torch.float32)
        self._upper_only = False     # a symmetric sweep left the blocks below the diagonal stale (mirrored on the next read of P)
        self._P_symmetric = True     # P is symmetric up to rounding (alpha * I and whatever the sweeps made of it); None: unchecked
        self._P_version = self._P._version   # (an in-place edit through the `P` property since the last sweep voids `_P_symmetric`)
        self.loss = 0.0
        self._backend = backend
        self._scratch = None
        self._partials = None
        self._colparts = None
        self.tiled = None  # None: by size; True / False: force the tiled / row-per-wave pass (tests, measurements)
        self.symmetric = None  # None: by size and updates per call; True / False: force / forbid the symmetric pass
        self.block = None      # None: with beta = 1: 8 (4 below 8 updates per call), else 1; 1: one pass over P per update; 2 / 4 / 8: one pass per so many updates
        self._blk = None

    # The symmetric pass streams only the blocks of P on and above the diagonal (4 n^2 instead of 8 n^2 bytes per update)
    # and mirrors them once per call: from a few updates per call on, on matrices far beyond the caches
    SYMMETRIC_ROWS = 32  # rows per tile of the symmetric pass (part of its summation order: rp_rls.sym_rows)
    SYMMETRIC_MIN_N = 8192
    SYMMETRIC_MIN_UPDATES = 4
    # updates per pass over P in the blocked symmetric sweep (rp_rls.block): 8 from 8 updates per call, 4 below.  A call of U
    # updates costs one read-only pass (0.5 ms at n_in = 32768), ceil(U / 8) - 1 full passes (0.9 ms) and one update-only
    # pass (0.9 ms) against 0.4 + 0.76 U ms with a pass per update: ahead from the 4 updates at which the symmetric pass starts
    BLOCK = 8
    BLOCK_ROWS = 128      # rows per tile of the blocked sweep (n_in = 32768, 8 updates per pass: 0.150 ms per update; 64 rows: 0.159, 32: 0.176)
    TILED_MIN_N = 8192  # measured (scripts/bench_rls.py, ms per update tiled / row-per-wave): n_in = 8192 0.108 / 0.113, 16384 0.386 / 0.401, 32768 1.43 / 1.68

    @property
    def P(self) -> torch.Tensor:
        """The state-error correlation matrix (reference edges.py:216).  The symmetric pass keeps only the 1024 x 1024 blocks
        on and above the diagonal up to date between sweeps; the blocks below are restored here, on the first read after
        such a sweep (one copy pass over half of P -- 0.78 ms at n_in = 32768 -- paid by whoever looks at P, not by every
        sweep)."""
        if self._upper_only:
            r = _abi.RpRls()
            r.n_in, r.ldp, r.P = self.n_in, self._P.stride(0), self._P.data_ptr()
            self._backend.rls_mirror(r, self._P.device)
            self._upper_only = False
        return self._P

    @P.setter
    def P(self, value: torch.Tensor):
        """A matrix assigned from outside is taken as it is.  The symmetric pass treats the blocks on and above the diagonal
        as authoritative, so it is only used on a P that is symmetric (up to the rounding the dense passes leave): that is
        checked once, at the next sweep that would take the symmetric pass (one comparison pass over P); an asymmetric P
        keeps the dense passes, which follow the reference's arithmetic on every element."""
        self._P = value
        self._upper_only = False
        self._P_symmetric = None  # unknown
        self._P_version = value._version

    def _symmetric_up_to_rounding(self) -> bool:
        """P == P^T up to the rounding the dense passes leave, checked in strips of 1024 rows (no full-matrix temporaries: at
        n_in = 32768 a whole-matrix `allclose` would allocate several times the 4 GiB of P)."""
        P, n = self._P, self.n_in
        scale = 0.0
        for a in range(0, n, 1024):
            scale = max(scale, float(P[a:a + 1024].abs().max()))
        for a in range(0, n, 1024):
            if not torch.allclose(P[a:a + 1024, a:], P[a:, a:a + 1024].T, rtol=1e-4, atol=1e-6 * scale):
                return False
        return True

    def to(self, device: str, **kwargs):
        self.weights = self.weights.detach().to(device).contiguous()
        sym = self._P_symmetric
        self.P = self.P.to(device).contiguous()
        self._P_symmetric = sym
        self._scratch = self._partials = self._colparts = self._blk = None   # (workspaces follow the device at the next sweep)
        return self

    def _ensure(self):
        if self._backend is None:
            from .engine import default_backend
            self._backend = default_backend()
        self._backend.check_device(self.weights.device)
        if self._scratch is None:
            dev = self.weights.device
            self._scratch = (torch.zeros(self.n_in, dtype=torch.float32, device=dev),
                             torch.zeros(self.n_in, dtype=torch.float32, device=dev),
                             torch.zeros(4 + self.n_out, dtype=torch.float32, device=dev))
            # tiled pass (P walked in address order by one-shot workgroups) once P is far beyond
            # the caches; below that the update is launch-bound and one wave per row is as fast
            tiles = (self.n_in + 1023) // 1024  # one partial per (row, 1024-column tile): rls_tile_kernel
            use = self.tiled is True or (self.tiled is None and self.n_in >= self.TILED_MIN_N)
            self._partials = torch.zeros(self.n_in * tiles, dtype=torch.float32, device=dev) \
                if use and self.n_in % 4 == 0 else None

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        """``weights @ x``.  The weights and P are float32 (the update runs in the gfx950 kernel); a float64 input -- the
        reference's default edge dtype -- is answered in float64."""
        w = self.weights if x.dtype == self.weights.dtype else self.weights.to(x.dtype)
        return x @ w.T if x.dim() == 2 else w @ x

    def sweep(self, X: torch.Tensor, Y: torch.Tensor, stride: int = 1):
        """Train over T consecutive steps: X [T, n_in] edge inputs, Y [T, n_out] targets, an update every `stride`
        steps (at 0, stride, ...).  Returns (pred [T, n_out], loss [T]) as the reference's loop would record them:
        the prediction of a step uses the weights in force before that step's update."""
        self._ensure()
        dev = self.weights.device
        T = int(X.shape[0])
        if T > 0 and (X.dim() != 2 or X.shape[1] != self.n_in):
            raise RuntimeError(f"edge input must be [T, {self.n_in}], got {tuple(X.shape)}")
        if Y.shape[0] != T or (T > 0 and Y.shape[1] != self.n_out):
            raise RuntimeError(f"targets must be [{T}, {self.n_out}], got {tuple(Y.shape)}")
        X = X.detach().to(dtype=torch.float32, device=dev).contiguous()
        Y = Y.detach().to(dtype=torch.float32, device=dev).contiguous()
        pred = torch.empty(T, self.n_out, dtype=torch.float32, device=dev)
        loss = torch.zeros(T, dtype=torch.float32, device=dev)
        if T == 0:
            return pred, loss
        z, zn, scratch = self._scratch
        r = _abi.RpRls()
        r.n_in, r.n_out = self.n_in, self.n_out
        r.ldp, r.ldw = self._P.stride(0), self.weights.stride(0)
        r.P, r.W = self._P.data_ptr(), self.weights.data_ptr()
        r.beta_inv = float(self.beta)
        r.z, r.zn, r.scratch = z.data_ptr(), zn.data_ptr(), scratch.data_ptr()
        r.partials = None if self._partials is None else self._partials.data_ptr()
        updates = (T + int(stride) - 1) // int(stride)
        use_sym = self._partials is not None and (
            self.symmetric is True or (self.symmetric is None and self.n_in >= self.SYMMETRIC_MIN_N
                                       and updates >= self.SYMMETRIC_MIN_UPDATES))
        if getattr(self, "_P_version", None) != self._P._version:
            # P was edited in place through the `P` property since the last sweep (the library itself writes through the raw
            # pointer and leaves the counter alone): what is known about its symmetry is void
            self._P_symmetric = None
        if use_sym and self._P_symmetric is None:
            # a P assigned or edited from outside: symmetric up to rounding (as the dense passes leave it: (k z_i) z_j and
            # (k z_j) z_i differ in the last bit)?  Then the half on and above the diagonal may stand for all of it.
            self._P_symmetric = self._symmetric_up_to_rounding()
        use_sym = use_sym and bool(self._P_symmetric)
        if self._upper_only and not use_sym:
            self.P  # (a dense pass reads all of P: restore the lower blocks first)
        if use_sym:
            block = self.block
            if block is None:
                # the default: blocked for the reference's own default beta = 1.  With beta < 1 its recursion makes P indefinite
                # and amplifies rounding whatever the form: there the literal one pass per update stays
                # the default and the blocked sweep is a choice (RLS.block = 8)
                block = (self.BLOCK if updates >= self.BLOCK else 4) if float(self.beta) == 1.0 else 1
            if block not in (1, 2, 4, 8):
                raise ValueError("RLS.block must be 1, 2, 4 or 8")
            rows = self.SYMMETRIC_ROWS if block == 1 else self.BLOCK_ROWS
            chunks = (self.n_in + rows - 1) // rows
            # column partials per chunk of `rows` rows (their sums per block row live in the z' kernel's LDS), per input of a block
            if self._colparts is None or self._colparts.shape[0] != block * chunks:
                self._colparts = torch.zeros(block * chunks, self.n_in, dtype=torch.float32, device=dev)
            tiles = (self.n_in + 1023) // 1024
            if self._partials.numel() < block * self.n_in * tiles:
                self._partials = torch.zeros(block * self.n_in * tiles, dtype=torch.float32, device=dev)
                r.partials = self._partials.data_ptr()
            r.colparts, r.ldc, r.sym_rows = self._colparts.data_ptr(), self._colparts.stride(0), rows
            r.defer_mirror = 1
            if block > 1:
                need = 2 * block * self.n_in + 4 * block * block + block * (1 + 2 * self.n_out)   # (rp_rls.blk)
                if self._blk is None or self._blk.numel() < need:
                    self._blk = torch.zeros(need, dtype=torch.float32, device=dev)
                r.block, r.blk = block, self._blk.data_ptr()
        self._backend.rls_sweep(r, T, int(stride), X.data_ptr(), X.stride(0) if T > 1 else self.n_in, Y.data_ptr(),
                                Y.stride(0) if T > 1 else self.n_out, pred.data_ptr(), self.n_out, loss.data_ptr(), dev)
        self._upper_only = bool(use_sym)
        self._P_version = self._P._version
        self.loss = loss[-1]
        return pred, loss

    def update(self, x: torch.Tensor, y: torch.Tensor, y_hat: Optional[torch.Tensor] = None) -> None:
        """One RLS step (signature of reference edges.py:227).  The weight update never depends on `y_hat`
        (edges.py:232); it only defines the recorded loss, so it is honoured when given."""
        _, loss = self.sweep(x.reshape(1, -1), torch.as_tensor(y).reshape(1, -1), 1)
        if y_hat is not None:
            err = torch.as_tensor(y).to(y_hat.device, torch.float32).reshape(-1) - y_hat.detach().reshape(-1)
            self.loss = torch.inner(err, err)
        else:
            self.loss = loss[0]
