"""G11-G13: the reference's SpikeResetNet (rectipy/nodes.py:333-401) stepping hand-written fields of
  qif_sfa  (neuron_model_templates/spiking_neurons/qif.yaml:25-35),
  lif      (neuron_model_templates/spiking_neurons/lif.yaml:10-23; input into I_ext and into s_ext),
  ik       (neuron_model_templates/spiking_neurons/ik.yaml:8-31).
Run in the build container only:  python tests/golden/make_golden_spiking_models.py"""
import numpy as np
import torch

from make_golden import import_reference, np_, save


def field_qif_sfa(n):
    def f(t, y, I_ext, spike, weights, tau, k, eta, tau_s, alpha, tau_x):
        v, s, x = y[:n], y[n:2 * n], y[2 * n:]
        dv = (v ** 2 + (eta - x) + I_ext) / tau + k * (weights @ s)
        ds = -s / tau_s + spike
        dx = -x / tau_x + alpha * spike
        return torch.cat((dv, ds, dx), 0)
    return f


def field_lif(n, synaptic_input):
    def f(t, y, ext, spike, weights, tau, k, eta, tau_s):
        v, s = y[:n], y[n:]
        I_ext = 0.0 if synaptic_input else ext
        s_ext = ext if synaptic_input else 0.0
        dv = -v / tau + k * (weights @ s) + I_ext + eta
        ds = -s / tau_s + spike + s_ext
        return torch.cat((dv, ds), 0)
    return f


def field_ik(n):
    def f(t, y, I_ext, spike, weights, C, k, v_r, v_theta, eta, g, E_r, b, tau_u, kappa, tau_s):
        v, u, s = y[:n], y[n:2 * n], y[2 * n:]
        dv = (k * (v - v_r) * (v - v_theta) - u + I_ext + eta + g * (weights @ s) * (E_r - v)) / C
        du = (b * (v - v_r) - u) / tau_u + kappa * spike
        ds = -s / tau_s + spike
        return torch.cat((dv, du, ds), 0)
    return f


def run(SpikeResetNet, field, n, n_state, y0, params, W, x, dt, theta, v_reset, reset_slice, out_slice, dtype, stride):
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    args = (t(y0), torch.zeros(x.shape[1], dtype=dtype), torch.zeros(n, dtype=dtype), t(W)) + tuple(t(p) for p in params)
    pm = {"in": 0, "spike_var": 1, "weights": 2}
    vm = {"out": out_slice, "reset_var": reset_slice}
    node = SpikeResetNet(field, args, vm, pm, spike_threshold=theta, spike_reset=v_reset, dt=dt, dtype=dtype, device="cpu")
    xs = torch.tensor(x, dtype=dtype)
    outs, events = [], []
    with torch.no_grad():
        for step in range(x.shape[0]):
            v_before = np_(node._y_spike).copy()
            outs.append(np_(node.forward(xs[step])).copy())
            for idx in np.nonzero(v_before >= theta)[0]:
                events.append((step, idx))
    final = np.concatenate([np_(node._y_start), np_(node._y_spike), np_(node._y_stop)])
    return np.stack(outs)[::stride], np.asarray(events, dtype=np.int32).reshape(-1, 2), final


def main():
    _, _, SpikeResetNet, _ = import_reference()
    dtypes = {"f64": torch.float64, "f32": torch.float32}

    # ---- G11 qif_sfa -------------------------------------------------------------------
    n, T, dt = 32, 4000, 1e-3
    r = np.random.default_rng(111)
    W = r.uniform(0, 1, (n, n)) * 20.0 / n
    eta = -5.0 + r.standard_normal(n)
    tau, k, tau_s = np.ones(n), np.ones(n), np.ones(n)
    alpha, tau_x = r.uniform(0.5, 1.5, n), r.uniform(5.0, 15.0, n)
    y0 = np.concatenate([np.full(n, -2.0), np.zeros(n), np.zeros(n)])
    x = np.full((T, 1), 15.0)
    out = {}
    for tag, dtype in dtypes.items():
        s, ev, fin = run(SpikeResetNet, field_qif_sfa(n), n, 3, y0, (tau, k, eta, tau_s, alpha, tau_x), W, x, dt, 100.0,
                         -100.0, (0, n), (n, 2 * n), dtype, 10)
        out.update({f"s_{tag}": s, f"events_{tag}": ev, f"final_{tag}": fin})
    save("g11_qif_sfa", dt=dt, W=W, eta=eta, tau=tau, k=k, tau_s=tau_s, alpha=alpha, tau_x=tau_x, y0=y0, x_const=15.0,
         theta=100.0, v_reset=-100.0, stride=10, T=T, **out)

    # ---- G12 lif: input into I_ext, and into s_ext --------------------------------------
    n, T, dt = 32, 3000, 1e-2
    r = np.random.default_rng(121)
    W = r.standard_normal((n, n)) * 2.0 / np.sqrt(n)
    tau, k, eta, tau_s = r.uniform(5, 10, n), r.uniform(0.5, 1.5, n), r.uniform(0.0, 0.5, n), r.uniform(0.3, 0.7, n)
    y0 = np.concatenate([r.uniform(0, 0.9, n), np.zeros(n)])
    x = 0.3 + 0.2 * r.standard_normal((T, n))
    out = {}
    for mode in ("membrane", "synapse"):
        for tag, dtype in dtypes.items():
            s, ev, fin = run(SpikeResetNet, field_lif(n, mode == "synapse"), n, 2, y0, (tau, k, eta, tau_s), W, x, dt, 1.0,
                             0.0, (0, n), (n, 2 * n), dtype, 10)
            out.update({f"s_{mode}_{tag}": s, f"events_{mode}_{tag}": ev, f"final_{mode}_{tag}": fin})
    save("g12_lif", dt=dt, W=W, tau=tau, k=k, eta=eta, tau_s=tau_s, y0=y0, x=x, theta=1.0, v_reset=0.0, stride=10, **out)

    # ---- G13 ik ---------------------------------------------------------------------------
    n, T, dt = 32, 4000, 1e-1
    r = np.random.default_rng(131)
    W = r.uniform(0, 1, (n, n)) * 15.0 / n
    P = dict(C=np.full(n, 100.0), k=np.full(n, 0.7), v_r=np.full(n, -60.0), v_theta=np.full(n, -40.0),
             eta=r.standard_normal(n) * 5.0, g=np.full(n, 1.0), E_r=np.zeros(n), b=np.full(n, -2.0),
             tau_u=np.full(n, 33.33), kappa=np.full(n, 10.0), tau_s=np.full(n, 6.0))
    order = ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_s")
    y0 = np.concatenate([np.full(n, -60.0) + r.standard_normal(n), np.zeros(n), np.zeros(n)])
    x = np.full((T, 1), 120.0)
    out = {}
    for tag, dtype in dtypes.items():
        s, ev, fin = run(SpikeResetNet, field_ik(n), n, 3, y0, tuple(P[q] for q in order), W, x, dt, 40.0, -55.0,
                         (0, n), (2 * n, 3 * n), dtype, 10)
        out.update({f"s_{tag}": s, f"events_{tag}": ev, f"final_{tag}": fin})
    save("g13_ik", dt=dt, W=W, y0=y0, x_const=120.0, theta=40.0, v_reset=-55.0, stride=10, T=T, **P, **out)


if __name__ == "__main__":
    main()
