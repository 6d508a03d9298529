"""Seam 2 of the reference: ``RateNet(rnn_func, rnn_args, var_map, param_map, ...)`` (rectipy/nodes.py:58-90) takes an
arbitrary Python callable ``rnn_func(t, y, *args) -> dy``.  A Python callable cannot run inside the fused gfx950 step
kernel, but the callables that reach this constructor in practice are the vector fields PyRates generates from the node
templates (or hand-written copies of them, rectipy_tests/test_nodes.py:32-33).  This module RECOGNISES such a callable:
it probes it on the host (float64, a handful of evaluations at construction time, never on the hot path) and, if it is
a member of the compiled leaky-integrator family

    dy = -y / tau + k * (W @ tanh(y)) + I_ext + eta          (leaky_integrator.yaml:10,33)

returns the per-neuron parameters the kernel needs and which entry of ``rnn_args`` plays which role.  Anything else is
refused with NotImplementedError -- there is no interpreter fallback.
"""
from typing import Callable, Dict, Optional, Tuple

import torch

_TOL = 1e-8


class FieldNotRecognised(NotImplementedError):
    pass


def _as64(a):
    return a.detach().to(dtype=torch.float64, device="cpu").clone() if isinstance(a, torch.Tensor) else a


def _solve_li(func: Callable, args: list, in_idx: int, W: torch.Tensor, n: int, gen: torch.Generator):
    """(tau, k, eta) of the leaky-integrator family from probes of `func`, or None if `func` is not in the family."""
    def F(y, I):
        a = list(args)
        a[in_idx] = I
        out = func(0, y, *a)
        if not isinstance(out, torch.Tensor) or out.shape != (n,):
            raise FieldNotRecognised("rnn_func does not map a length-N state to a length-N derivative")
        return out.to(torch.float64)

    zero = torch.zeros(n, dtype=torch.float64)
    eta = F(zero, zero)
    e = torch.randn(n, generator=gen, dtype=torch.float64)
    scale = 1.0 + float(eta.abs().max())
    if not bool(torch.isfinite(eta).all()):
        return None
    if not float((F(zero, e) - eta - e).abs().max()) <= _TOL * scale:  # the input must enter additively, one-to-one
        return None
    for _ in range(4):
        y1, y2 = (torch.randn(n, generator=gen, dtype=torch.float64) for _ in range(2))
        c1, c2 = W @ torch.tanh(y1), W @ torch.tanh(y2)
        g1, g2 = F(y1, zero) - eta, F(y2, zero) - eta
        det = -y1 * c2 + y2 * c1
        if float(det.abs().min()) > 1e-6:
            break
    else:
        return None
    a = (g1 * c2 - g2 * c1) / det          # 1 / tau
    k = (-y1 * g2 + y2 * g1) / det
    y3, i3 = (torch.randn(n, generator=gen, dtype=torch.float64) for _ in range(2))
    pred = -y3 * a + k * (W @ torch.tanh(y3)) + i3 + eta
    got = F(y3, i3)
    if not float((got - pred).abs().max()) <= 1e-7 * (1.0 + float(got.abs().max())):  # (NaN-safe comparison)
        return None
    if not float(a.abs().min()) >= 1e-12:
        return None  # no leak term: tau would be infinite
    return 1.0 / a, k, eta


def recognise_rate_field(func: Callable, rnn_args: tuple, param_map: Dict[str, int]
                         ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, str]]:
    """Returns (W [N, N] float64, {"tau","k","eta"} per-neuron float64 vectors, roles {param_map name -> engine
    name}).  Raises FieldNotRecognised when `func` is not the tanh leaky-integrator field."""
    y0 = rnn_args[0]
    n = int(y0.shape[0])
    args = [_as64(a) for a in rnn_args[1:]]
    if "in" not in param_map:
        raise KeyError("param_map needs the index of the input slot ('in')")
    in_idx = int(param_map["in"])
    w_idx: Optional[int] = param_map.get("weights")
    if w_idx is None:
        cands = [i for i, a in enumerate(args) if isinstance(a, torch.Tensor) and a.dim() == 2 and tuple(a.shape) == (n, n)]
        if len(cands) != 1:
            raise FieldNotRecognised("cannot tell which entry of rnn_args is the N x N coupling matrix; name it in "
                                     "param_map['weights']")
        w_idx = cands[0]
    W = args[w_idx]
    if not isinstance(W, torch.Tensor) or tuple(W.shape) != (n, n):
        raise FieldNotRecognised(f"rnn_args[{w_idx + 1}] is not an {n} x {n} coupling matrix")
    gen = torch.Generator().manual_seed(1234)
    try:
        base = _solve_li(func, args, in_idx, W, n, gen)
    except FieldNotRecognised:
        raise
    except Exception as err:  # the callable itself failed on a probe (wrong arity, shape error, ...)
        raise FieldNotRecognised(f"rnn_func could not be evaluated on the probe states: {err}") from err
    if base is None:
        raise FieldNotRecognised(
            "rnn_func is not the tanh leaky-integrator field dy = -y/tau + k*(W @ tanh(y)) + I_ext + eta; arbitrary "
            "Python vector fields cannot run in the fused kernel (use RateNet.from_pyrates with a built-in template)")
    tau, k, eta = base
    # which named argument is which parameter: perturb it and see which identified parameter follows
    roles = {}
    for name, idx in param_map.items():
        if name in ("in", "weights") or idx == in_idx or idx == w_idx:
            continue
        old = args[idx]
        if not isinstance(old, (torch.Tensor, float, int)):
            continue
        new = old * 1.25 + 0.5
        trial = list(args)
        trial[idx] = new
        try:
            res = _solve_li(func, trial, in_idx, W, n, torch.Generator().manual_seed(99))
        except Exception:
            res = None
        if res is None:
            raise FieldNotRecognised(f"parameter '{name}' does not act as tau, k or eta of the leaky-integrator field")
        expect = new if isinstance(new, torch.Tensor) else torch.full((n,), float(new), dtype=torch.float64)
        hits = [eng for eng, val, ref in (("tau", res[0], tau), ("k", res[1], k), ("eta", res[2], eta))
                if float((val - ref).abs().max()) > 1e-9]
        if len(hits) != 1 or float(({"tau": res[0], "k": res[1], "eta": res[2]}[hits[0]] - expect).abs().max()) > 1e-6:
            raise FieldNotRecognised(f"parameter '{name}' does not act as tau, k or eta of the leaky-integrator field")
        roles[name] = hits[0]
    return W, {"tau": tau, "k": k, "eta": eta}, roles
