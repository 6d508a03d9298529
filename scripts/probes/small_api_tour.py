"""us per step of the front end's entry points at the reference's own scale (N = 500): run (with and without recorded variables),
fit_bptt, fit_rls, fit_ridge, test -- tanh and QIF -- host time against wall time (a gap = the host ran ahead, no gap = the host or
a hidden synchronisation bounds the call)."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import Network  # noqa: E402

dev = "cuda:0"
N, T, m, k = 500, 4000, 3, 2
P = "neuron_model_templates."
rng = np.random.default_rng(0)
x = rng.standard_normal((T, m)).astype(np.float32)
y = (rng.standard_normal((T, k)) * 0.1).astype(np.float32)


def build(kind, train=None, readout_train=None):
    with contextlib.redirect_stdout(io.StringIO()):
        net = Network(dt=1e-2 if kind == "tanh" else 1e-3, device=dev)
        net.add_func_node("inp", m, "identity")
        if kind == "tanh":
            net.add_diffeq_node("rnn", P + "rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext", output_var="tanh_op/r",
                                weights=rng.standard_normal((N, N)) * 1.2 / N ** 0.5, source_var="tanh_op/r", target_var="li_op/r_in",
                                dtype=torch.float32, train_params=train)
        else:
            net.add_diffeq_node("rnn", P + "spiking_neurons.qif.qif", input_var="qif_op/I_ext", output_var="qif_op/s",
                                weights=np.abs(rng.standard_normal((N, N))) * 2.0 / N, source_var="qif_op/s", target_var="qif_op/s_in",
                                spike_var="qif_op/spike", reset_var="qif_op/v", spike_threshold=100.0, spike_reset=-100.0,
                                node_vars={"all/qif_op/eta": rng.uniform(0.5, 3.0, N)}, dtype=torch.float32, train_params=train)
        net.add_func_node("out", k, "identity")
        net.add_edge("inp", "rnn", weights=rng.standard_normal((N, m)).astype(np.float32))
        net.add_edge("rnn", "out", weights=(rng.standard_normal((k, N)) / N ** 0.5).astype(np.float32), train=readout_train)
    return net


def timed(name, fn, steps):
    with contextlib.redirect_stdout(io.StringIO()):
        fn()   # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(f"{name:58s} {wall / steps * 1e6:8.2f} us per step (host {host / steps * 1e6:7.2f})", flush=True)


for kind in ("tanh", "qif"):
    net = build(kind)
    timed(f"{kind}: run, every step sampled", lambda: net.run(x, sampling_steps=1, verbose=False, enable_grad=False), T)
    timed(f"{kind}: run, sampling_steps=10", lambda: net.run(x, sampling_steps=10, verbose=False, enable_grad=False), T)
    var = "li_op/v" if kind == "tanh" else "qif_op/v"
    timed(f"{kind}: run, sampling_steps=10, one recorded variable",
          lambda: net.run(x, sampling_steps=10, verbose=False, enable_grad=False, record_vars=[("rnn", var, False)]), T)
    timed(f"{kind}: test (sampling_steps=100)", lambda: net.test(x, y[::100], sampling_steps=100, verbose=False), T)
    net = build(kind, train=["weights"], readout_train="gd")
    timed(f"{kind}: fit_bptt, sgd, windows of 100",
          lambda: net.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=100, sampling_steps=100, verbose=False), T)
    timed(f"{kind}: fit_bptt, adam, windows of 100",
          lambda: net.fit_bptt(x, y, optimizer="adam", lr=1e-4, update_steps=100, sampling_steps=100, verbose=False), T)
    net = build(kind, readout_train="rls")
    timed(f"{kind}: fit_rls, an update every step", lambda: net.fit_rls(x, y, update_steps=1, sampling_steps=100, verbose=False), T)
    net = build(kind)
    timed(f"{kind}: fit_ridge (sampling_steps=10)", lambda: net.fit_ridge(x, y, sampling_steps=10, verbose=False, add_readout_node=False), T)
