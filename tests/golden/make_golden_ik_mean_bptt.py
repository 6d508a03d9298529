"""G20-G21: surrogate-gradient BPTT through the reference's SpikeResetNet for the population-mean fields iku_op and
ik_biexp_op (ik.yaml:34-68): one SGD window of Network.fit_bptt, every parameter trained (updates = -lr * gradient).
Run in the build container only:  python tests/golden/make_golden_ik_mean_bptt.py"""
import numpy as np

from make_golden import import_reference
from make_golden_ik_mean import field_ik_biexp, field_iku
from make_golden_spiking_bptt import run_case


def main():
    Network, _, SpikeResetNet, _ = import_reference()
    n, T, dt, kout = 24, 200, 1e-1, 2
    for name, field, extra, order, n_state, seed in (
            ("g20_iku_bptt", field_iku(n), ("tau_s",),
             ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_s"), 3, 201),
            ("g21_ik_biexp_bptt", field_ik_biexp(n), ("tau_r", "tau_d"),
             ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_r", "tau_d"), 4, 211)):
        r = np.random.default_rng(seed)
        W = r.uniform(0, 1, (n, n)) * 15.0 / n
        P = dict(C=r.uniform(90, 110, n), k=r.uniform(0.6, 0.8, n), v_r=r.uniform(-62, -58, n),
                 v_theta=r.uniform(-42, -38, n), eta=r.standard_normal(n) * 5.0, g=r.uniform(0.8, 1.2, n),
                 E_r=r.uniform(-2, 2, n), b=r.uniform(-2.5, -1.5, n), tau_u=r.uniform(30, 36, n),
                 kappa=r.uniform(8, 12, n))
        if "tau_s" in extra:
            P["tau_s"] = r.uniform(5, 7, n)
        else:
            P["tau_r"], P["tau_d"] = r.uniform(1.5, 2.5, n), r.uniform(5, 7, n)
        y0 = np.concatenate([r.uniform(-60.0, 35.0, n), r.uniform(0, 20, n), r.uniform(0, 0.5, n)] +
                            ([r.uniform(0, 0.2, n)] if n_state == 4 else []))
        x = 120.0 + 5.0 * r.standard_normal((T, n))
        run_case(Network, SpikeResetNet, name, field, n, y0, order, P, W, x, r.standard_normal((T, kout)) * 0.1,
                 r.standard_normal((kout, n)) / np.sqrt(n), dt, 40.0, -55.0, 1e-3, (2 * n, 3 * n))


if __name__ == "__main__":
    main()
