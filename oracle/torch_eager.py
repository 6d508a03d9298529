"""CPU baseline: the stock PyTorch eager per-step loop that is op-for-op what the reference's
``Network.run`` / ``RateNet.forward`` execute for a tanh rate population (reference
rectipy/network.py:588-599 calling rectipy/nodes.py:166-170 with the field of
leaky_integrator.yaml:10,33, hand form rectipy_tests/test_nodes.py:32-33) -- a "port" of the
reference's CPU path, used only by bench.py's ``cpu_baseline`` leg and by tests.

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by rectipy_amd.
"""
import time

import torch


def li_tanh_field(t, y, I_ext, weights, tau, k, eta):
    return -y / tau + k * (weights @ torch.tanh(y)) + I_ext + eta


def run_eager(W, tau, k, eta, y, x, dt, steps, sampling_steps=1, cutoff=0):
    """`steps` forward steps with the reference's buffering/sampling; returns (samples, y)."""
    samples, buffer = [], []
    with torch.no_grad():
        for step in range(steps):
            y_old = y
            y = y_old + dt * li_tanh_field(0, y_old, x[step, :], W, tau, k, eta)
            out = y_old
            if step >= cutoff:
                buffer.append(out)
                if step % sampling_steps == 0:
                    samples.append(torch.mean(torch.stack(buffer, dim=0), dim=0))
                    buffer = []
    return samples, y


def time_eager(W, tau, k, eta, y, x, dt, warmup, steps):
    """steps/s of the eager loop on the host cores torch is configured to use."""
    _, y = run_eager(W, tau, k, eta, y, x, dt, warmup)
    t0 = time.perf_counter()
    run_eager(W, tau, k, eta, y, x[warmup:], dt, steps)
    return steps / (time.perf_counter() - t0)
