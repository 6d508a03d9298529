"""Differential fuzz, build container only: random SEQUENCES of calls on a QIF SpikeResetNet inside a Network --
run with / without gradient tracking, reset (whole state), a write into net.state (which the reference ignores), surrogate-
gradient fit_bptt windows -- the reference's own SpikeResetNet (hand-written QIF field, float64) against
rectipy_amd (CPU oracle backend, float32).  Thresholds are low (theta = 1, v_reset = -1) so that every sequence spikes.
Compared after every call: the output trajectory, the lagging state y, trained parameters.  Spike trains must agree
exactly as long as no membrane potential passes the threshold within the float32 noise of the two runs."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import import_reference, make_qif  # noqa: E402

RefNetwork, _, RefSpikeResetNet, RefSpike = import_reference()
from oracle.backend import OracleBackend  # noqa: E402
import rectipy_amd  # noqa: E402

QIF = "neuron_model_templates.spiking_neurons.qif.qif"
PM = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}


def build(rng, n, k, train, dt, theta, v_reset):
    W = np.abs(rng.standard_normal((n, n))) * 2.0 / n
    tau, kk, eta, tau_s = rng.uniform(0.8, 1.2, n), rng.uniform(0.5, 1.5, n), rng.uniform(-1.0, 0.5, n), rng.uniform(0.5, 2.0, n)
    y0 = np.concatenate([rng.uniform(-0.9, 0.9, n), rng.uniform(0, 0.2, n)])
    W_out = rng.standard_normal((k, n)) / np.sqrt(n)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    args = (t(y0), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64), t(W), t(tau), t(kk), t(eta), t(tau_s))
    vm = {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n), "qif_op/s": (n, 2 * n)}
    node = RefSpikeResetNet(make_qif(n), args, vm, dict(PM), spike_threshold=theta, spike_reset=v_reset, dt=dt,
                            dtype=torch.float64, device="cpu", train_params=train)
    ref = RefNetwork(dt, dtype=torch.float64)
    ref.add_node("qif", node, node_type="diff_eq", op="qif_op")
    ref.add_func_node("out", k, "identity")
    ref.add_edge("qif", "out", weights=W_out.copy(), train="gd", dtype=torch.float64)
    amd = rectipy_amd.Network(dt, device="cpu", backend=OracleBackend())
    amd.add_diffeq_node("qif", QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=W, source_var="qif_op/s",
                        target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v", train_params=train,
                        dtype=torch.float32, spike_threshold=theta, spike_reset=v_reset,
                        node_vars={"all/qif_op/tau": tau, "all/qif_op/k": kk, "all/qif_op/eta": eta, "all/qif_op/tau_s": tau_s,
                                   "all/qif_op/v": y0[:n], "all/qif_op/s": y0[n:]})
    amd.add_func_node("out", k, "identity")
    amd.add_edge("qif", "out", weights=W_out.copy(), train="gd")
    return ref, amd


def main(cases=30, seed=0):
    rng = np.random.default_rng(seed)
    worst, worst_at, spikes_total = 0.0, None, 0
    for case in range(cases):
        n, k = int(rng.integers(3, 24)), int(rng.integers(1, 3))
        train = [None, ["weights"], ["qif_op/eta", "qif_op/tau"], ["weights", "qif_op/k", "qif_op/tau_s"]][int(rng.integers(0, 4))]
        dt, theta, v_reset = 5e-3, 1.0, -1.0
        ref, amd = build(rng, n, k, train, dt, theta, v_reset)
        log = []
        for op_i in range(int(rng.integers(2, 6))):
            op = ["run", "run_nograd", "write", "reset", "fit"][int(rng.integers(0, 5))]
            log.append(op)
            dev = 0.0
            if op in ("run", "run_nograd"):
                T = int(rng.integers(20, 120))
                x = np.full((T, n), 3.0) + rng.standard_normal((T, n)) * 0.5
                kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                          enable_grad=op == "run")
                a = ref.run(x, **kw).to_numpy("out")
                b = amd.run(x.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.size else 0.0
            elif op == "write":
                with torch.no_grad():
                    ref.state["qif"][:] = 0.25
                    amd.state["qif"][:] = 0.25
                ref.detach()  # (the reference cannot backpropagate through a y that was written in place)
                amd.detach()
            elif op == "reset":
                y = np.concatenate([rng.uniform(-0.9, 0.9, n), rng.uniform(0, 0.2, n)])
                ref.reset({"qif": y})
                amd.reset({"qif": y})
            else:
                w = int(rng.integers(10, 40))
                T = w * int(rng.integers(1, 3))
                x = np.full((T, n), 3.0) + rng.standard_normal((T, n)) * 0.5
                tgt = rng.standard_normal((T, k)) * 0.1
                kw = dict(optimizer="sgd", lr=1e-3, update_steps=w, sampling_steps=1, verbose=False)
                with contextlib.redirect_stdout(io.StringIO()):
                    a = ref.fit_bptt(x, tgt, **kw).to_numpy("out")
                    b = amd.fit_bptt(x.astype(np.float32), tgt.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max())
                dev = max(dev, float(np.abs(ref.get_edge("qif", "out").weights.detach().numpy()
                                            - amd.get_edge("qif", "out").weights.detach().numpy()).max()))
                if train and "weights" in train:
                    dev = max(dev, float(np.abs(ref.get_node("qif")._args[PM["weights"]].detach().numpy()
                                                - amd.get_node("qif")["weights"].detach().numpy()).max()))
            if op != "write":  # (after a write both sides hold 0.25 in the lagging copy; the integration state is untouched)
                ya, yb = ref.state["qif"].detach().numpy(), amd.state["qif"].detach().numpy()
                # v of a neuron that has just been reset differs by the reset jump when a spike lands a step apart: compare s
                dev = max(dev, float(np.abs(ya[n:] - yb[n:]).max()))
            if dev > worst:
                worst, worst_at = dev, (case, list(log), train, n)
            if dev > 2e-3:
                print("MISMATCH", dev, case, log, train, n)
                return 1
    print(f"{cases} random QIF call sequences: worst deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
