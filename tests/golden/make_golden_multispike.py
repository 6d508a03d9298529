"""G29: the reference's MultiSpikeResetNet (rectipy/nodes.py:404-465) -- two QIF populations (exc / inh) in one node,
hand-constructed like G4 (the PyRates circuit compiler is not available): state vector [v_e, s_e, v_i, s_i], two
spike inputs, two reset slices, block coupling.  Captures the class's particular timing: POST-update output, spike
detection on a private copy that starts at zero and survives reset().
Run in the build container only:  python tests/golden/make_golden_multispike.py"""
import numpy as np
import torch

from make_golden import import_reference, np_, save


def main():
    import_reference()
    from rectipy.nodes import MultiSpikeResetNet
    ne, ni, T, dt = 24, 8, 3000, 1e-3
    n = ne + ni
    r = np.random.default_rng(290)
    W = np.zeros((n, n))
    W[:, :ne] = r.uniform(0, 1, (n, ne)) * 30.0 / ne     # excitatory columns
    W[:, ne:] = -r.uniform(0, 1, (n, ni)) * 20.0 / ni    # inhibitory columns
    eta = np.concatenate([-4.0 + r.standard_normal(ne), -6.0 + r.standard_normal(ni)])
    tau = np.concatenate([np.full(ne, 1.0), np.full(ni, 0.8)])
    tau_s = np.concatenate([np.full(ne, 1.0), np.full(ni, 0.5)])
    k = np.ones(n)
    v0 = np.concatenate([r.uniform(-3, 2, ne), r.uniform(-3, 2, ni)])
    v0[3] = 150.0  # above threshold at t = 0: the zero-initialised detector must NOT spike it on the first step
    s0 = np.zeros(n)
    y0 = np.concatenate([v0[:ne], s0[:ne], v0[ne:], s0[ne:]])
    ve, se, vi, si = (0, ne), (ne, 2 * ne), (2 * ne, 2 * ne + ni), (2 * ne + ni, 2 * n)

    def field(t, y, I_ext, spike_e, spike_i, W, tau, k, eta, tau_s):
        v = torch.cat((y[ve[0]:ve[1]], y[vi[0]:vi[1]]))
        s = torch.cat((y[se[0]:se[1]], y[si[0]:si[1]]))
        spike = torch.cat((spike_e, spike_i))
        dv = (v ** 2 + eta + I_ext) / tau + k * (W @ s)
        ds = -s / tau_s + spike
        return torch.cat((dv[:ne], ds[:ne], dv[ne:], ds[ne:]))

    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(y0), torch.zeros(1, dtype=dtype), torch.zeros(ne, dtype=dtype), torch.zeros(ni, dtype=dtype), t(W), t(tau),
                t(k), t(eta), t(tau_s))
        pm = {"in": 0, "spike_var_0": 1, "spike_var_1": 2, "weights": 3}
        vm = {"out": se, "spike_reset_0": ve, "spike_reset_1": vi, "e/qif_op/v": ve, "e/qif_op/s": se, "i/qif_op/v": vi,
              "i/qif_op/s": si}
        node = MultiSpikeResetNet(field, args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                                  device="cpu")
        x = torch.tensor([12.0], dtype=dtype)
        outs, ys = [], []
        with torch.no_grad():
            for step in range(T):
                o = np_(node.forward(x)).copy()
                if step < 5 or step % 10 == 0:
                    outs.append(o)
                    ys.append(np_(node.y).copy())
            out["out_" + tag], out["y_" + tag] = np.stack(outs), np.stack(ys)
            out["y_end_" + tag] = np_(node.y).copy()
            # reset(): y is replaced, the spike detector is not -> the next step detects on the OLD reset variables
            y1 = y0.copy()
            y1[ve[0] + 5] = 130.0  # above threshold in the new state, below in the stale detector: no spike
            det = np.concatenate([np_(node._y_reset[0]), np_(node._y_reset[1])])
            node.reset(y1)
            after = []
            for step in range(3):
                after.append(np_(node.forward(x)).copy())
            out["after_reset_" + tag] = np.stack(after)
            out["y_after_reset_" + tag] = np_(node.y).copy()
            out["detector_before_reset_" + tag] = det
        n_sp = int((np.diff(out["y_" + tag][:, se[0]:se[1]], axis=0) > 0.5).sum())
        print(tag, "jumps of s_e seen in the stored rows:", n_sp, "max s_e", out["y_" + tag][:, se[0]:se[1]].max())
    kept = np.array([s for s in range(T) if s < 5 or s % 10 == 0])
    save("g29_multispike", dt=dt, ne=ne, ni=ni, W=W, eta=eta, tau=tau, tau_s=tau_s, k=k, v0=v0, s0=s0, y0=y0, y1=y1,
         x_const=12.0, theta=100.0, v_reset=-100.0, kept=kept, **out)


if __name__ == "__main__":
    main()
