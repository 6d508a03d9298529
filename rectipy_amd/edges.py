"""Static / gradient-trained linear projections between nodes (reference ``Linear``,
rectipy/edges.py:8-65).  Input and readout projections are k x N with k << N: they are not
on the HBM-bound path.  Inside the fused forward window the input projection is evaluated in
the step kernel's epilogue (RP_X_PROJ); everywhere else (batched readout of a sampled
trajectory, autograd for trained edges) it is a plain torch matmul.
"""
from typing import Iterator, Union

import numpy as np
import torch


class Linear:
    """``weights @ x`` with weights of shape [n_out, n_in]."""

    _tensors = ["weights"]

    def __init__(self, n_in: int, n_out: int, weights: Union[np.ndarray, torch.Tensor] = None,
                 dtype: torch.dtype = torch.float64, detach: bool = True, **kwargs):
        # (dtype defaults to float64 as in the reference, edges.py:12; Network.add_edge passes the network's float32)
        if weights is None:
            weights = torch.randn(n_out, n_in, dtype=dtype)
        elif not isinstance(weights, torch.Tensor):
            weights = torch.tensor(np.asarray(weights), dtype=dtype)
        else:
            weights = weights.detach().to(dtype).clone()
        if weights.dim() != 2:
            raise ValueError("edge weights must be a matrix")
        # the reference accepts either orientation and transposes whatever matches (n_in, n_out) FIRST (edges.py:22-26)
        # -- a square matrix therefore IS transposed: edge weights are read as [source, target] when ambiguous
        # (verified against the reference's class; fixture G27).
        if weights.shape[0] == n_in and weights.shape[1] == n_out:
            weights = weights.T.contiguous()
        elif weights.shape[0] != n_out or weights.shape[1] != n_in:
            raise ValueError("Shape of the provided weights does not match the input and output dimensions of the "
                             "source and target nodes.")
        self.n_in = n_in
        self.n_out = n_out
        self.weights = weights
        self.train_params = []
        if not detach:
            self.weights.requires_grad_(True)
            self.train_params.append(self.weights)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        """x: [n_in] (one step, as the reference) or [T, n_in] (a whole window)."""
        if x.dim() == 2:
            return x @ self.weights.T
        return self.weights @ x

    def parameters(self, recurse: bool = True) -> Iterator:
        for p in self.train_params:
            yield p

    def to(self, device: str, **kwargs):
        req = self.weights.requires_grad
        w = self.weights.detach().to(device).requires_grad_(req)
        if req:
            self.train_params = [w]
        self.weights = w
        return self

    def detach(self):
        pass
