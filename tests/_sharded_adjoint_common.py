"""Shared by tests/test_sharded_adjoint.py and its rank processes: the synthetic window and the one-process mirror of the
sharded adjoint sweep."""
import torch

from rectipy_amd import _abi
from rectipy_amd.engine import Population


def window_data(N, T, model, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, N, generator=g) * 1.2 / N ** 0.5
    params = dict(tau=torch.rand(N, generator=g) * 2 + 1, k=torch.rand(N, generator=g) + 0.5, eta=torch.randn(N, generator=g) * 0.1)
    v0 = torch.randn(N, generator=g) * 0.5
    x = torch.randn(T, N, generator=g) * 0.5
    kw = {}
    if model == "qif":
        W = W.abs() * 2
        params = dict(tau=torch.ones(N), k=torch.ones(N), eta=params["eta"] + 2.0, tau_s=torch.rand(N, generator=g) + 0.5)
        kw = dict(theta=4.0, v_reset=-4.0)
        x = x + 10.0
    g_out = torch.randn(T, N, generator=g) * 0.1
    return W, params, v0, x, g_out, kw


def shard_population(model, N, world, rank, dev, backend, W, params, v0, kw):
    rows = N // world
    sl = slice(rank * rows, (rank + 1) * rows)
    extra = dict(kw)
    if model == "qif":
        extra["s0"] = torch.zeros(rows)
    pop = Population(model, W[sl], {k: v[sl] for k, v in params.items()}, 1e-2, dev, v0=v0[sl], n=N, row0=rank * rows,
                     backend=backend, **extra)
    return pop, sl


def backward_kwargs(model, pop, T, dev, saved, g_out_local, x_local):
    """Buffers of one rank's adjoint sweep (updated in place by it): returns (kwargs of Population.backward, results)."""
    rows = pop.rows
    z = lambda *shape: torch.zeros(*shape, device=dev)
    res = dict(lam=z(rows), u_traj=z(T, rows), grad_tau=z(rows), grad_k=z(rows), grad_eta=z(rows))
    kw = dict(traj_v=saved["v"], traj_r=saved["r"], grad_out=g_out_local, grad_var=_abi.RP_OUT_SOURCE, lam=res["lam"],
              u_traj=res["u_traj"], traj_acc=saved["acc"], grad_tau=res["grad_tau"], grad_k=res["grad_k"], grad_eta=res["grad_eta"])
    if model == "qif":
        res["lam_s"], res["grad_tau_s"] = z(rows), z(rows)
        kw.update(lam_s=res["lam_s"], grad_tau_s=res["grad_tau_s"], slope=0.5, x=x_local)
    return kw, res
