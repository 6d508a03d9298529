"""G18-G19: the reference's SpikeResetNet stepping the hand-written iku_op and ik_biexp_op fields
(neuron_model_templates/spiking_neurons/ik.yaml:34-68; mean(.) = mean over the N neurons).
Run in the build container only:  python tests/golden/make_golden_ik_mean.py"""
import numpy as np
import torch

from make_golden import import_reference, save
from make_golden_spiking_models import run


def field_iku(n):
    def f(t, y, I_ext, spike, weights, C, k, v_r, v_theta, eta, g, E_r, b, tau_u, kappa, tau_s):
        v, u, s = y[:n], y[n:2 * n], y[2 * n:]
        dv = (k * (v - v_r) * (v - v_theta) - u + I_ext + eta + g * (weights @ s) * (E_r - v)) / C
        du = (b * (torch.mean(v) - v_r) - u) / tau_u + kappa * torch.mean(spike)
        ds = -s / tau_s + spike
        return torch.cat((dv, du, ds), 0)
    return f


def field_ik_biexp(n):
    def f(t, y, I_ext, spike, weights, C, k, v_r, v_theta, eta, g, E_r, b, tau_u, kappa, tau_r, tau_d):
        v, u, s, x = y[:n], y[n:2 * n], y[2 * n:3 * n], y[3 * n:]
        dv = (k * (v - v_r) * (v - v_theta) - u + I_ext + eta + g * (weights @ s) * (E_r - v)) / C
        du = (b * (torch.mean(v) - v_r) - u) / tau_u + kappa * torch.mean(spike)
        ds = -s / tau_d + x
        dx = -x / tau_r + spike
        return torch.cat((dv, du, ds, dx), 0)
    return f


def main():
    _, _, SpikeResetNet, _ = import_reference()
    n, T, dt = 32, 4000, 1e-1
    r = np.random.default_rng(181)
    W = r.uniform(0, 1, (n, n)) * 15.0 / n
    base = dict(C=np.full(n, 100.0), k=np.full(n, 0.7), v_r=np.full(n, -60.0), v_theta=np.full(n, -40.0),
                eta=r.standard_normal(n) * 5.0, g=np.full(n, 1.0), E_r=np.zeros(n), b=np.full(n, -2.0),
                tau_u=np.full(n, 33.33), kappa=np.full(n, 10.0))
    x = np.full((T, 1), 120.0)
    for name, field, extra, order, n_state, out_slice in (
            ("g18_iku", field_iku(n), dict(tau_s=np.full(n, 6.0)),
             ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_s"), 3, (2 * n, 3 * n)),
            ("g19_ik_biexp", field_ik_biexp(n), dict(tau_r=np.full(n, 2.0), tau_d=np.full(n, 6.0)),
             ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_r", "tau_d"), 4, (2 * n, 3 * n))):
        P = dict(base, **extra)
        y0 = np.concatenate([np.full(n, -60.0) + r.standard_normal(n)] + [np.zeros(n)] * (n_state - 1))
        out = {}
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            s, ev, fin = run(SpikeResetNet, field, n, n_state, y0, tuple(P[q] for q in order), W, x, dt, 40.0, -55.0,
                             (0, n), out_slice, dtype, 10)
            out.update({f"s_{tag}": s, f"events_{tag}": ev, f"final_{tag}": fin})
        save(name, dt=dt, W=W, y0=y0, x_const=120.0, theta=40.0, v_reset=-55.0, stride=10, T=T, **P, **out)


if __name__ == "__main__":
    main()
