"""Reservoir computing at scale: a fixed random reservoir (p = 0.1, as in the reference's documentation/ridge_example.py)
of N tanh rate neurons classifies which of three input sequences is being played; only the read-out is fitted, by ridge
regression on the collected states.  Because W is not trained it is streamed in its packed sparse form (bit-identical to
the dense product, `rectipy_amd/sparse.py`).

    python examples/reservoir_ridge.py [N]          (needs the MI355X; default N = 16384)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from anywhere without PYTHONPATH
import json
import sys
import time

import numpy as np
import torch

from rectipy_amd import Network, input_connections, random_connectivity, wta_score

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p, J, tau, m, dt = 0.1, 0.9, 0.5, 5, 1e-2   # echo-state regime (J < 1), neurons faster than a syllable
signals = ([0, 2, 1], [4, 2, 3], [3, 2, 0])
k, syll_steps, n_reps = len(signals), 50, 40
rng = np.random.default_rng(0)


def epoch():
    steps = n_reps * 3 * syll_steps
    inp, tgt = np.zeros((steps, m), dtype=np.float32), np.zeros((steps, k), dtype=np.float32)
    for rep in range(n_reps):
        choice = rng.integers(k)
        for idx, channel in enumerate(signals[choice]):
            a = (rep * 3 + idx) * syll_steps
            inp[a:a + syll_steps, channel] = 1.0
        tgt[rep * 3 * syll_steps:(rep + 1) * 3 * syll_steps, choice] = 1.0
    return inp, tgt


W = random_connectivity(N, N, p, normalize=True).astype(np.float32) * J
W_in = input_connections(N, m, 0.5, variance=2.0, zero_mean=True).astype(np.float32)

t0 = time.perf_counter()
net = Network(dt=dt, device="cuda:0")
node = net.add_diffeq_node("tanh", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", weights=W,
                           source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                           output_var="li_op/v", dtype=torch.float32, node_vars={"all/li_op/tau": tau})
net.add_func_node("inp", m, "identity")
net.add_edge("inp", "tanh", weights=W_in)
net.run(np.zeros((2000, m), dtype=np.float32), verbose=False, sampling_steps=2001)        # wash out the initial state
torch.cuda.synchronize()
t_build = time.perf_counter() - t0

coeffs, t_fit, steps_done = [], 0.0, 0
for _ in range(3):
    inp, tgt = epoch()
    t0 = time.perf_counter()
    obs = net.fit_ridge(inp, tgt, sampling_steps=5, alpha=1e-2, verbose=False, add_readout_node=False)
    torch.cuda.synchronize()
    t_fit += time.perf_counter() - t0
    steps_done += inp.shape[0]
    coeffs.append(obs["w_out"].detach().cpu().numpy())
net.add_func_node("readout", k, "identity")
net.add_edge("tanh", "readout", weights=np.mean(coeffs, axis=0).T)
inp, tgt = epoch()
obs, loss = net.test(inp, tgt, loss="mse", sampling_steps=1, verbose=False)
print(json.dumps({"N": N, "packed_sparse": node._pop.packed is not None,
                  "stored_MB": None if node._pop.packed is None else round(node._pop.packed.stored_bytes / 1e6, 1),
                  "build_and_washout_s": round(t_build, 2), "fit_ridge_s": round(t_fit, 2),
                  "fit_steps_per_s": round(steps_done / t_fit, 1), "test_mse": round(float(loss), 4),
                  "wta_score": round(float(wta_score(obs.to_numpy("out"), tgt)), 3)}))
