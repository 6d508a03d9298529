"""Recording container returned by ``Network.run`` / ``fit_bptt`` / ``test`` -- the data API of
the reference's ``Observer`` (rectipy/observer.py:9-128): ``obs["out"]``, ``obs[(node, var)]``,
``obs["steps"]``, ``obs["loss"]``, ``to_numpy``, ``to_dataframe``, ``save``.

The fused GPU window writes all samples of a run into one preallocated ``[S, n]`` device buffer;
``record_block`` stores such blocks without copying and they are unbound into the per-sample
list the reference exposes only when somebody asks for it.  Plotting is out of scope.
"""
from typing import Any, Iterable, Tuple, Union

import numpy as np
import torch


class Observer:

    def __init__(self, dt: float, record_output: bool = True, record_loss: bool = True, record_vars: list = None):
        record_vars = record_vars or []
        self._dt = dt
        self._state_vars = [tuple(v[:2]) for v in record_vars]
        self._reduce_vars = [bool(v[2]) if len(v) > 2 else False for v in record_vars]
        self._record_loss = record_loss
        self._record_out = record_output
        self._keys = list(self._state_vars)
        if record_loss:
            self._keys.append("loss")
        if record_output:
            self._keys.append("out")
        self._keys.append("steps")
        # per key: list of chunks; a chunk is either a list of per-sample items or a [S, ...] tensor block
        self._chunks = {k: [] for k in self._keys}
        self._additional_storage = {}

    # ---- reference data API -------------------------------------------------------------
    def __getitem__(self, item: Union[str, Tuple[str, str]]):
        if item in self._chunks:
            return self._flat(item)
        return self._additional_storage[item]

    @property
    def recorded_state_variables(self) -> list:
        return self._state_vars

    @property
    def recorded_variables(self) -> list:
        return list(self._keys)

    @property
    def recordings(self):
        """All recorded quantities as one DataFrame indexed by time (reference observer.py:62-70)."""
        from pandas import DataFrame
        columns = list(self._state_vars) + (["out"] if self._record_out else []) + (["loss"] if self._record_loss else [])
        data = {str(c): list(self.to_numpy(c)) for c in columns}
        return DataFrame(index=np.asarray(self["steps"]) * self._dt, data=data)

    def record(self, step: int, output: torch.Tensor, loss: Union[float, torch.Tensor],
               record_vars: Iterable[torch.Tensor]) -> None:
        """One sample (reference observer.py:79-105)."""
        self._append("steps", [step])
        for key, val, reduce in zip(self._state_vars, record_vars, self._reduce_vars):
            # state vectors are updated in place by the step kernels: store a copy, not a view of the live state
            self._append(key, [torch.mean(val) if reduce else val.detach().clone()])
        if self._record_out:
            self._append("out", [output])
        if self._record_loss:
            self._append("loss", [loss])

    def record_block(self, steps, outputs: torch.Tensor, loss, record_vars: Iterable[torch.Tensor]) -> None:
        """S samples at once: ``outputs`` is [S, n_out], each recorded variable [S, n]."""
        steps = [int(s) for s in steps]
        if not steps:
            return
        self._chunks["steps"].append(steps)
        for key, val, reduce in zip(self._state_vars, record_vars, self._reduce_vars):
            self._chunks[key].append(val.mean(dim=1) if reduce else val)
        if self._record_out:
            self._chunks["out"].append(outputs)
        if self._record_loss:
            self._chunks["loss"].append([loss] * len(steps) if not isinstance(loss, (list, tuple)) else list(loss))

    def save(self, key: str, val: Any):
        self._additional_storage[key] = val

    def to_numpy(self, item: Union[str, Tuple[str, str]]) -> np.ndarray:
        if item == "loss" and item in self._chunks:
            return np.asarray(self._flat("loss"), dtype=np.float64)
        if item in self._chunks:
            parts = []
            for c in self._chunks[item]:
                if isinstance(c, torch.Tensor):
                    parts.append(c.detach().cpu().numpy())
                elif len(c):
                    parts.append(np.asarray([v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for v in c]))
            return np.concatenate(parts, axis=0) if parts else np.zeros((0,))
        val = self._additional_storage[item]
        return np.asarray([v.detach().cpu().numpy() for v in val])

    def to_dataframe(self, item: Union[str, Tuple[str, str]]):
        from pandas import DataFrame
        try:
            data = self.to_numpy(item)
            return DataFrame(index=np.asarray(self["steps"]) * self._dt, data=data)
        except (KeyError, AttributeError):
            return self[item]

    def stacked(self, item) -> torch.Tensor:
        """All samples of `item` as one [S, ...] tensor (what the reference builds with torch.stack(obs[item]))."""
        parts = []
        for c in self._chunks[item]:
            parts.append(c if isinstance(c, torch.Tensor) else torch.stack([torch.as_tensor(v) for v in c]))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

    # ---- thin matplotlib conveniences (same call signatures as reference observer.py:130-198) ------------
    @staticmethod
    def _axis_label(v) -> str:
        return f"Node: {v[0]}, variable: {v[-1]}" if isinstance(v, tuple) else str(v)

    @staticmethod
    def _axes(ax, kwargs):
        if ax is None:
            import matplotlib.pyplot as plt
            figsize = kwargs.pop("figsize", None)
            _, ax = plt.subplots(**({"figsize": figsize} if figsize else {}))
        return ax

    def _series(self, item) -> np.ndarray:
        val = self[item]
        if isinstance(val, (list, tuple)):
            return np.asarray([v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for v in val])
        return np.asarray(val)

    def plot(self, y, x=None, ax=None, **kwargs):
        """Line plot of recording `y` against time (or against recording / stored item `x`)."""
        ax = self._axes(ax, kwargs)
        ys = self._series(y)
        xs = np.asarray(self["steps"]) * self._dt if x is None else self._series(x)
        ax.plot(xs[:len(ys)] if len(xs) >= len(ys) else np.arange(len(ys)), ys, **kwargs)
        ax.set_xlabel("time" if x is None else self._axis_label(x))
        ax.set_ylabel(self._axis_label(y))
        return ax

    def matshow(self, v, ax=None, **kwargs):
        """Colour plot (neurons x time) of recording `v`."""
        import matplotlib.pyplot as plt
        ax = self._axes(ax, kwargs)
        shrink = kwargs.pop("shrink", 0.6)
        im = ax.imshow(self._series(v).T, **kwargs)
        plt.colorbar(im, ax=ax, shrink=shrink)
        ax.set_xlabel("time")
        ax.set_ylabel(self._axis_label(v))
        return ax

    # ---- internals ------------------------------------------------------------------------
    def _append(self, key, items):
        chunks = self._chunks[key]
        if chunks and isinstance(chunks[-1], list):
            chunks[-1].extend(items)
        else:
            chunks.append(list(items))

    def _flat(self, key) -> list:
        out = []
        for c in self._chunks[key]:
            out.extend(c.unbind(0) if isinstance(c, torch.Tensor) else c)
        if key == "loss":  # losses recorded lazily as 0-dim device tensors: one transfer, python floats as the reference
            lazy = [i for i, v in enumerate(out) if isinstance(v, torch.Tensor)]
            if lazy:
                vals = torch.stack([out[i].detach().reshape(()) for i in lazy]).cpu().tolist()
                for i, v in zip(lazy, vals):
                    out[i] = v
        return out
