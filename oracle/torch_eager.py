"""CPU baseline: the stock PyTorch eager per-step loop that is op-for-op what the reference's
``Network.run`` / ``RateNet.forward`` execute for a tanh rate population (reference
rectipy/network.py:588-599 calling rectipy/nodes.py:166-170 with the field of
leaky_integrator.yaml:10,33, hand form rectipy_tests/test_nodes.py:32-33) -- a "port" of the
reference's CPU path, used only by bench.py's ``cpu_baseline`` leg and by tests.

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by rectipy_amd.
"""
import time

import torch


def numa_nodes() -> int:
    """NUMA nodes of the host (1 when the sysfs tree is absent)."""
    import glob
    return max(1, len(glob.glob("/sys/devices/system/node/node[0-9]*")))


def cpu_share() -> float:
    """CPUs this process may use at once: the cgroup quota (cpu.max, v2; cfs_quota_us / cfs_period_us, v1) when one is
    set, else the affinity mask's size.  A GPU box hands a one-GPU job a SHARE of a 256-thread host; more threads than
    that only time-slice."""
    import os
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except AttributeError:
        return float(os.cpu_count() or 1)


def host_copy_spread(t: torch.Tensor) -> torch.Tensor:
    """Host copy of `t` whose pages are FIRST TOUCHED by all hardware threads in parallel, so that a large matrix is
    spread over the memory of every NUMA node instead of sitting on the node of the thread that called ``.cpu()``
    (a 4.3 GB coupling matrix on one node caps the DRAM-bound mat-vec at that node's bandwidth, whatever the thread
    count).  torch's fill and copy kernels split their range statically over the intra-op pool, as the BLAS mat-vec
    splits its rows."""
    import os
    default = torch.get_num_threads()
    try:
        torch.set_num_threads(os.cpu_count() or default)
        out = torch.empty(t.shape, dtype=t.dtype)
        out.fill_(0)                      # parallel first touch
        out.copy_(t.detach().cpu())       # parallel copy from the staging tensor
    finally:
        torch.set_num_threads(default)
    return out


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


def sweep_threads(fn, candidates):
    """Run ``fn()`` (returns steps/s) under each intra-op thread count; returns (best_threads, best, table).
    The default thread count (all hardware threads) is rarely the best for a DRAM-bound mat-vec."""
    default = torch.get_num_threads()
    table = {}
    try:
        for t in candidates:
            torch.set_num_threads(int(t))
            table[int(t)] = fn()
    finally:
        torch.set_num_threads(default)
    best = max(table, key=table.get)
    return best, table[best], table


def time_eager_bptt(W, tau, k, eta, y, x, W_out, target, dt, steps):
    """One BPTT window of `steps` steps the way the reference runs it on the CPU: the per-step loop under autograd
    (recurrent weights trained), `MSELoss` on the stacked read-out predictions, `backward()`
    (reference rectipy/network.py:1028-1040, 1123-1130).  Returns steps/s of forward + backward."""
    W = W.clone().requires_grad_(True)
    y = y.clone().requires_grad_(True)
    loss_fn = torch.nn.MSELoss()
    t0 = time.perf_counter()
    preds = []
    for step in range(steps):
        y_old = y
        y = y_old + dt * li_tanh_field(0, y_old, x[step, :], W, tau, k, eta)
        preds.append(W_out @ y_old)
    loss = loss_fn(torch.stack(preds), target[:steps])
    loss.backward()
    return steps / (time.perf_counter() - t0)


def qif_field(n):
    def qif(t, y, I_ext, spike, weights, tau, k, eta, tau_s):  # qif.yaml:11-12
        v, s = y[:n], y[n:]
        return torch.cat(((v ** 2 + eta + I_ext) / tau + k * (weights @ s), -s / tau_s + spike), 0)
    return qif


def time_eager_qif(W, tau, k, eta, tau_s, v, s, x_const, dt, theta, v_reset, warmup, steps):
    """steps/s of the reference's spiking step on the CPU (rectipy/nodes.py:382-392: spike detection on the old v,
    `spike / dt` input, Euler on the concatenated state, reset blend), op for op, no_grad."""
    n = v.shape[0]
    f = qif_field(n)
    x = torch.tensor([x_const], dtype=v.dtype)
    t0 = None
    with torch.no_grad():
        for step in range(warmup + steps):
            if step == warmup:
                t0 = time.perf_counter()
            spikes = torch.heaviside(v - theta, torch.ones((), dtype=v.dtype))
            y = torch.cat((v, s), 0)
            y_new = y + dt * f(0, y, x, spikes / dt, W, tau, k, eta, tau_s)
            reset = spikes.detach()
            v = y_new[:n] * (1.0 - reset) + reset * v_reset
            s = y_new[n:]
    return steps / (time.perf_counter() - t0)


def time_eager_rls(n_in, n_out, updates, beta=0.99, alpha=1.0, seed=0):
    """updates/s of the reference's RLS.update on the CPU, op for op (rectipy/edges.py:227-234): z = beta^-1 P x,
    k = 1 / (1 + x.z), the weight update through two outer products, P -= k outer(z, z) (an n_in x n_in temporary),
    loss.  float32, P spread over the NUMA nodes by a parallel first touch."""
    import os
    g = torch.Generator().manual_seed(seed)
    default = torch.get_num_threads()
    torch.set_num_threads(os.cpu_count() or default)
    P = torch.empty(n_in, n_in)
    P.fill_(0)
    torch.set_num_threads(default)
    P.diagonal().fill_(alpha)
    W = torch.zeros(n_out, n_in)
    beta_inv = 1.0 / beta
    X = torch.randn(updates + 1, n_in, generator=g) * 0.1
    Y = torch.randn(updates + 1, n_out, generator=g)
    t0 = None
    with torch.no_grad():
        for t in range(updates + 1):
            if t == 1:
                t0 = time.perf_counter()
            x, y = X[t], Y[t]
            y_hat = W @ x
            z = beta_inv * P @ x
            k = (1.0 + x @ z) ** (-1)
            error = y - y_hat
            W += torch.outer((y - k * x @ (W + torch.outer(y, z)).T), z)
            P -= k * torch.outer(z, z)
            _ = torch.inner(error, error)
    return updates / (time.perf_counter() - t0)
