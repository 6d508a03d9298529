"""G35: SpikeResetNet + QIF by the reference at N = 2052 (8 full 256-column chunks + a ragged 4-column chunk, several row
groups per workgroup), T = 2500 steps of the BASELINE configs[3] recipe (W = rand * 20 / N, eta = -5 + N(0, 1), constant
input 15, dt = 1e-3, threshold 100 / reset -100; rectipy/nodes.py:382-392): every spike (step, neuron) event, s every 10th
step on every 8th neuron, final state -- in float64 AND float32 (the events must agree between the two for the fixture to
be usable as an exact reference; the script says so).

Run in the build container only:  python tests/golden/make_golden_qif_n2052.py   (a few minutes)"""
import numpy as np
import torch

import recipes
from make_golden import import_reference, make_qif, np_, save
from make_golden_large import fp


def main():
    _, _, SpikeResetNet, _ = import_reference()
    n, T, dt = 2052, 2500, 1e-3
    P = recipes.qif_network(n, seed=3500)
    cols = np.arange(0, n, 8)
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(P["y0"]), torch.zeros(1, dtype=dtype), torch.zeros(n, dtype=dtype), t(P["W"]), t(P["tau"]), t(P["k"]),
                t(P["eta"]), t(P["tau_s"]))
        pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
        vm = {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n), "qif_op/s": (n, 2 * n)}
        node = SpikeResetNet(make_qif(n), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                             device="cpu")
        xin = torch.tensor([15.0], dtype=dtype)
        outs, events = [], []
        with torch.no_grad():
            for step in range(T):
                v_before = np_(node._y_spike).copy()
                o = np_(node.forward(xin))
                if step % 10 == 0:
                    outs.append(o[cols].copy())
                for idx in np.nonzero(v_before >= 100.0)[0]:
                    events.append((step, idx))
        out["s_" + tag] = np.stack(outs).astype(np.float32)
        out["events_" + tag] = np.asarray(events, dtype=np.int32).reshape(-1, 2)
        out["v_final_" + tag] = np_(node._y_spike)
        out["s_final_" + tag] = np_(node._y_stop)
        print(f"G35 {tag}: {len(events)} spikes")
    same = out["events_f64"].shape == out["events_f32"].shape and bool((out["events_f64"] == out["events_f32"]).all())
    print("G35: f32 and f64 reference spike events identical:", same)
    save("g35_qif_n2052", dt=dt, n=n, T=T, seed=3500, x_const=15.0, theta=100.0, v_reset=-100.0, stride=10, cols=cols,
         events_agree=same, **fp(W=P["W"], eta=P["eta"]), **out)


if __name__ == "__main__":
    main()
