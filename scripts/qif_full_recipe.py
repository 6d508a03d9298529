"""BASELINE.json configs[3] exactly as stated, through the front end: QIF population, N = 65536,
W = rand * 20 / N, tau = k = tau_s = 1, eta = -5, v0 = -2, s0 = 0, theta = 100, v_reset = -100, dt = 1e-3, constant input
15, 20000 steps, Network.run with sampling_steps = 100.  (bench.py times 400 steps of the same population from a spread
start, because the per-step cost is stationary; this script is the full recipe, ~50 s on one MI355X.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rectipy_amd import Network

dev = torch.device("cuda:0")
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.empty(N, N, device=dev)
for r in range(0, N, 4096):
    W[r:r + 4096] = torch.rand(4096, N, device=dev, generator=gen)
W.mul_(20.0 / N)
net = Network(dt=1e-3, device="cuda:0")
net.add_diffeq_node("qif", "neuron_model_templates.spiking_neurons.qif.qif", input_var="qif_op/I_ext", output_var="qif_op/s",
                    weights=W, source_var="qif_op/s", target_var="qif_op/s_in", spike_var="qif_op/spike",
                    reset_var="qif_op/v", spike_threshold=100.0, spike_reset=-100.0, dtype=torch.float32,
                    node_vars={"all/qif_op/eta": -5.0, "all/qif_op/tau": 1.0, "all/qif_op/k": 1.0, "all/qif_op/tau_s": 1.0,
                               "all/qif_op/v": -2.0, "all/qif_op/s": 0.0})
del W
node = net.get_node("qif")
node.enable_spike_recording(1 << 24)
x = torch.full((T, 1), 15.0, device=dev)
net.run(x[:50], sampling_steps=50, verbose=False, enable_grad=False)   # warm-up (also measures the walk of W)
net.reset()
node.enable_spike_recording(1 << 24)
torch.cuda.synchronize()
t0 = time.perf_counter()
obs = net.run(x, sampling_steps=100, verbose=False, enable_grad=False)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ev = node.spikes()
s_mean = obs.to_numpy("out").mean(axis=1)
print(json.dumps({"workload": f"QIF N={N}, {T} steps from v0=-2 (BASELINE.json configs[3] as stated), Network.run, sampling_steps=100",
                  "steps_per_s": T / wall, "ms_per_step": wall / T * 1e3,
                  "achieved_GBps": (4.0 * N * N + 4.0 * 12 * N) / (wall / T) / 1e9,
                  "frac_of_8TBps": (4.0 * N * N + 4.0 * 12 * N) / (wall / T) / 1e9 / 8000,
                  "spikes": int(ev.shape[0]), "first_spike_step": int(ev[:, 0].min()) if ev.shape[0] else None,
                  "spikes_per_neuron_per_unit_time": ev.shape[0] / N / (T * 1e-3),
                  "mean_s_first_last_sample": [float(s_mean[0]), float(s_mean[-1])], "samples": len(obs["steps"]),
                  "state_finite": bool(torch.isfinite(node.current_state).all().item())}))
