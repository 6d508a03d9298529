#!/usr/bin/env python3
"""Headline benchmark: integration steps/s of an N-neuron dense tanh rate RNN on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--neurons 32768]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` (N > 1) without a launcher starts the N ranks itself: the parent touches no GPU, spawns
`python -m torch.distributed.run` on a free port as a CHILD process, relays rank 0's JSON line and returns the
children's exit status (`self_launch`).

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): N = 32768 neurons,
W = 1.5 * randn(N, N) / sqrt(N) (generated on device in 1024-row blocks seeded by block index, so
every sharding sees the same matrix), tau = 10, k = 1, eta = 0, v(0) = 0, dt = 1e-2, a broadcast
sinusoidal input, sampling_steps = 100.  A "step" is one explicit-Euler step of the whole network
(= one launch of the fused step kernel).  Inputs are resident in HBM when the timed region starts.

With --gpus N > 1 the SAME network is row-sharded over N ranks (strong scaling): each rank steps its
rows, then the ranks all-gather the new source vector over RCCL/xGMI, every step.

The JSON line carries `roofline` (algorithmic bytes per launch / measured launch duration, against
the 8 TB/s HBM peak) and, at N = 1, `cpu_baseline` (a torch-eager port of the reference's CPU loop,
timed on this box's host cores for a bounded number of steps) plus `extra` with the fwd+bwd window.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def step_kernel_label(streaming):
    """The step kernel instance the library launched: its streaming configuration comes from the library itself
    (rp_forward_tuning: the static rule, or the one rp_tune_forward measured for this shape on this device)."""
    return (f"step_kernel<LI_TANH,R={streaming['rows_per_wave']},U={streaming['chunks_in_flight']},vec4,nt> on "
            f"{streaming['workgroups'] or 'rows/4R'} workgroups ({'measured' if streaming['measured'] else 'static rule'})")


def gen_rows(n, row0, rows, device, g=1.5, block=1024):
    """Rows [row0, row0+rows) of the synthetic coupling matrix; identical for every sharding."""
    W = torch.empty(rows, n, dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    r = row0
    while r < row0 + rows:
        b0 = (r // block) * block
        gen.manual_seed(1000 + b0 // block)
        blk = torch.randn(min(block, n - b0), n, dtype=torch.float32, device=device, generator=gen)
        lo, hi = r - b0, min(b0 + block, row0 + rows) - b0
        W[r - row0:r - row0 + (hi - lo)] = blk[lo:hi]
        r = b0 + hi
    W.mul_(g / n ** 0.5)
    return W


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks, argv, script=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run (one
    node, rendezvous on 127.0.0.1 at a free port) as CHILDREN of this process, which has not touched the GPU and never
    replaces itself.  Rank 0's JSON line is relayed to stdout as the only stdout line; everything else the ranks print
    goes to stderr.  Returns the children's exit status (non-zero when any rank failed or no JSON line appeared)."""
    import subprocess
    script = script or os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    print("[bench] launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        print(f"[bench] the {n_ranks}-rank job did not finish within {timeout} s and was killed", file=sys.stderr)
        sys.stderr.write(out or "")
        return 124
    for ln in (out or "").splitlines():
        try:
            rec = json.loads(ln)
        except ValueError:
            rec = None
        if isinstance(rec, dict) and "metric" in rec and "value" in rec:
            lines.append(ln)
        elif ln.strip():
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"[bench] torch.distributed.run exited with status {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if len(lines) != 1:
        print(f"[bench] expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--neurons", type=int, default=32768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (RCCL) path even with one rank")
    ap.add_argument("--exchange", choices=("rccl", "direct"), default=os.environ.get("RECTIPY_AMD_EXCHANGE", "rccl"),
                    help="per-step exchange of the row-sharded run: the RCCL all-gather enqueued in-library (default), or the "
                         "stream-ordered direct push exchange (RP_COMM_DIRECT; opt-in: rehearsed with two processes on one "
                         "GPU only, never over xGMI)")
    ap.add_argument("--config5-neurons", type=int, default=262144,
                    help="N of the row-sharded BASELINE configs[4] leg that runs with --gpus > 1 (0: skip)")
    ap.add_argument("--config5-steps", type=int, default=200)
    ap.add_argument("--qif-steps", type=int, default=20000,
                    help="Euler steps of the QIF leg (BASELINE configs[3] states 20000: about 49 s on one MI355X)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become the launcher (nothing above this line has touched the GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    # stdout carries the ONE JSON line and nothing else: libraries that print to the C-level stdout (RCCL's version banner
    # at communicator creation, hipBLASLt notices) are sent to stderr for the whole run; the line is written to the saved
    # descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str):
        os.write(json_fd, (line + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        from rectipy_amd.sharded import one_rank_env
        one_rank_env()  # --force-dist without a launcher: fills in only the missing variables, free port
        dist.init_process_group("nccl", device_id=dev)

    N, K, Wm = args.neurons, args.steps, args.warmup
    dt = 1e-2
    T_total = Wm + K
    tt = torch.arange(max(T_total, 3000), device=dev, dtype=torch.float32) * dt
    x = torch.sin(2 * torch.pi * 0.2 * tt).reshape(-1, 1).contiguous()
    sampling = 100
    n_samples = T_total // sampling + 2
    params = dict(tau=10.0, k=1.0, eta=0.0)
    net = sp = None

    if not use_dist:
        # single GPU: through the reference-compatible front end (Network.add_diffeq_node / Network.run)
        from rectipy_amd import Network
        row0, rows = 0, N
        net = Network(dt=dt, device=str(dev))
        net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh",
                            input_var="li_op/I_ext", output_var="li_op/v", weights=gen_rows(N, 0, N, dev),
                            source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32,
                            node_vars={"all/li_op/tau": 10.0, "all/li_op/k": 1.0, "all/li_op/eta": 0.0})
        pop = net.get_node("rnn")._pop

        def run(steps, step0, idx):
            obs = net.run(x[step0:step0 + steps], sampling_steps=sampling, verbose=False, enable_grad=False)
            return len(obs["steps"])
    else:
        from rectipy_amd.sharded import ShardedPopulation, shard_bounds
        row0, rows = shard_bounds(N, world, rank)
        sp = ShardedPopulation("li_tanh", gen_rows(N, row0, rows, dev), params, dt, dev, n=N, exchange=args.exchange)
        pop = sp.pop
        samples = torch.zeros(n_samples, rows, device=dev)
        exchange = "torch.distributed all_gather_into_tensor per step (host loop)"
        had_comm = sp.comm is not None
        # collective (every rank calls it, whether or not it got a communicator): the in-library exchange must be
        # available on ALL ranks and reproduce the torch.distributed one bit for bit; otherwise every rank drops it
        if sp.check_inlib_exchange(3):
            exchange = ("stream-ordered direct push exchange (RP_COMM_DIRECT: peer stores + hipStreamWrite/WaitValue32)"
                        if sp.direct is not None else "RCCL all-gather enqueued in-library (rp_forward_sharded)") + \
                "; checked == torch.distributed path"
        elif had_comm:
            exchange += " [in-library exchange unavailable on some rank or disagreed with it: disabled]"

        def run(steps, step0, idx):
            return sp.forward(steps, x=x[step0:step0 + steps], step0=step0, sampling_steps=sampling,
                              samples=samples, sample_idx=idx)

    def barrier():
        if use_dist:
            dist.barrier()

    # Everything the harness itself needs is set up BEFORE the warm-up steps, so that nothing but the contract's
    # synchronize + barrier lies between them and the timed region: a GPU that idles for 10 ms or more runs its next
    # kernels 1-3 % slower while the clocks come back (measured: 20 steps after a 0 / 1 / 10 / 100 ms gap average
    # 607 / 607 / 615 / 626 us per step), and a full Python garbage collection takes that long.
    import gc
    gc.collect()
    gc.disable()  # (a collection inside a 12 ms timed region would be a visible fraction of it)
    # HIP events on the launch stream, recorded immediately around the library call that enqueues the K step kernels
    # (not around the front end's host-side preparation, during which the stream is idle): average launch duration,
    # including the gaps between consecutive launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # (events are created on their first record: not inside the timed region)
    e1.record()
    backend, calls, stamps = pop.backend, [], []
    entry = "forward_sharded" if use_dist and sp.comm is not None else "forward"
    inner = getattr(backend, entry)
    timing = [False]

    def bracketed(*a, **kw):
        if not timing[0]:
            return inner(*a, **kw)
        if not calls:
            stamps.append(time.perf_counter())
            e0.record()
        calls.append(1)
        inner(*a, **kw)
        e1.record()
        stamps.append(time.perf_counter())

    setattr(backend, entry, bracketed)
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    try:
        idx = run(Wm, 0, 0)  # the process's FIRST call of the path: its wall time is reported as first_call.cold_ms
    except BaseException:
        delattr(backend, entry)
        raise
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    barrier()
    torch.cuda.synchronize()
    timing[0] = True
    t0 = time.perf_counter()
    try:
        idx = run(K, Wm, idx)
    finally:
        delattr(backend, entry)  # (the instance attribute shadowed the method)
    t_ret = time.perf_counter()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gc.enable()
    if not calls:
        raise RuntimeError("the timed region never reached the library's forward entry point")
    kernel_ms = e0.elapsed_time(e1) / K
    if use_dist:
        tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    finite = bool(torch.isfinite(pop.v).all().item())
    streaming = pop.backend.forward_tuning(pop._net())

    sharded_bwd = None
    if use_dist and "--exchange" in sys.argv:
        try:
            sharded_bwd = sharded_fwd_bwd_leg(sp, x, world, rank, dev)
        except Exception as e:  # (reported, never fatal: the forward line above is the contract)
            sharded_bwd = {"error": f"{type(e).__name__}: {e}"}
    config5 = None
    if use_dist and args.config5_neurons > 0:
        sp = pop = samples = None  # release the strong-scaling shard first
        torch.cuda.empty_cache()
        config5 = config5_leg(args.config5_neurons, args.config5_steps, world, rank, dev, dt)

    if rank != 0:
        dist.destroy_process_group()
        return

    steps_per_s = K / wall
    # algorithmic bytes per launch: W once + per-neuron vectors (v r/w, tau, k, eta, r r/w) + input
    algo_bytes = 4.0 * rows * N + 4.0 * (7 * rows + 1)
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters need their own rocprofv3 passes (scripts/archive/profile_kernels.sh); the
    # figure is read from the committed summary of that run and labelled as such -- it is NOT measured by this run
    traffic, traffic_source = None, None
    for name in ("r05_step_kernel_traffic.json", "r04_step_kernel_traffic.json", "r03_step_kernel_traffic.json", "r02_step_kernel_traffic.json",
                 "r01_step_kernel_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if world == 1 and N == 32768 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_source = (f"profiles/{name} (separate rocprofv3 --pmc passes of the same kernel and shape; "
                              "not measured in this run)")
            break
    out = {
        "metric": "integration steps/sec (fwd), N-neuron dense tanh rate RNN",
        "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"tanh rate RNN N={N} dense W fp32, forward-only Euler, dt=1e-2, sampling_steps=100, "
                               f"broadcast sinusoid input (BASELINE.json configs[1])",
                   "neurons": N, "rows_per_gpu": rows, "parallelism": f"row-shard x{world}; " + exchange
                   if use_dist else "single GPU, driven through rectipy_amd.Network.run",
                   "rccl_ranks": dist.get_world_size() if use_dist else 1,
                   "exchange": exchange if use_dist else "none (one GPU holds every row)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": step_kernel_label(streaming), "algorithmic_bytes_per_launch": algo_bytes,
                     "avg_launch_ms": kernel_ms},
        "state_finite": finite,
        # where the wall clock of the timed region went on rank 0: the front end returned (all launches enqueued), the
        # GPU finished, barrier + second synchronize done
        "timed_region_ms": {"library_entered": (stamps[0] - t0) * 1e3, "library_returned": (stamps[-1] - t0) * 1e3,
                            "front_end_returned": (t_ret - t0) * 1e3, "gpu_done": (t_gpu - t0) * 1e3, "total": wall * 1e3,
                            "step_kernels": kernel_ms * K},
    }

    if use_dist:
        out["roofline"]["note"] = ("row-sharded run: avg_launch_ms is the per-step time on rank 0's stream, i.e. the shard's "
                                   "step kernel PLUS the in-stream all-gather; achieved/frac therefore understate the kernel")
    if config5 is not None:
        out.setdefault("extra", {})["config5"] = config5
    if sharded_bwd is not None:
        out.setdefault("extra", {})["sharded_fwd_bwd"] = sharded_bwd
    cpu_threads = None
    if world == 1 and not use_dist and not args.no_cpu_baseline:
        out["cpu_baseline"], cpu_threads = cpu_baseline(pop, x, dt, N)
    if world == 1 and not use_dist and not args.no_extra:
        out["extra"] = fwd_bwd_extra(pop, x, N, dev)
        from rectipy_amd import tuning
        # cold vs warm cost of one call through the front end: the W-step warm-up call was this process's first call of
        # the path (library load, code-object load, first launches; NO walk measurement under the default policy), the
        # timed call a warm one; "overhead" = wall time minus steps x the measured launch duration
        out["extra"]["first_call"] = {
            "steps": Wm, "cold_ms": first_call_ms, "cold_overhead_ms": first_call_ms - Wm * kernel_ms,
            "warm_steps": K, "warm_ms": wall * 1e3, "warm_overhead_ms": wall * 1e3 - K * kernel_ms,
            "autotune_policy": tuning.mode(), "walk_measured_in_this_process": bool(streaming["measured"])}
        if cpu_threads is not None:
            out["extra"]["cpu_fwd_bwd"] = cpu_fwd_bwd(pop, x, dt, N, cpu_threads)
        out["extra"]["batched_trajectories"] = batched_extra(pop, N, dev, out["ms_per_step"], net=net)
        out["extra"]["fit_ridge"] = ridge_extra(net, x, N, dev)
        out["extra"]["fit_ridge"]["phases_ms"]["step_kernels_alone_at_the_measured_launch_time"] = 2000 * kernel_ms
        out["extra"].update(facade_fit_bptt(pop, x, N, dev, teacher_net=net))  # it trains (modifies) the shared weight matrix
        pop = net = sp = None  # release the 4.3 GB network before the two larger side workloads
        torch.cuda.empty_cache()
        out["extra"]["stepwise"] = stepwise_extra(dev)
        out["extra"]["small_n"] = small_n_extra(dev)
        out["extra"]["rls_n32768"] = rls_extra(dev, cpu_threads=cpu_threads)
        out["extra"]["qif_n65536"] = qif_extra(dev, T=args.qif_steps, cpu_threads=cpu_threads)
        out["extra"]["sparse_p0.1_n32768"] = sparse_extra(dev)
        torch.cuda.empty_cache()
        if os.environ.get("RECTIPY_AMD_BENCH_NO_CHILD") != "1":  # (skipped under a profiler: its preload would follow the children)
            out["extra"]["one_rank_exchange"] = one_rank_exchange_extra()
    emit(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def fwd_bwd_extra(pop, x, N, dev):
    """BASELINE.json configs[2]: BPTT over a 1000-step window (forward with trajectory capture, adjoint
    sweep, MFMA dW GEMM), MSE on a k=3 readout.  Timed with events on the launch stream."""
    from rectipy_amd import _abi
    T = 1000
    f32 = dict(dtype=torch.float32, device=dev)
    tv, tr, ut = (torch.empty(T, N, **f32) for _ in range(3))
    gen = torch.Generator(device=dev).manual_seed(2)
    W_out = torch.randn(3, N, generator=gen, **f32) / N ** 0.5
    target = torch.randn(T, 3, generator=gen, **f32) * 0.1
    grad = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def window(Tw):
        v_, r_, u_ = tv[:Tw], tr[:Tw], ut[:Tw]
        ev[0].record()
        pop.forward(Tw, x=x[:Tw], save_v=v_, save_src=r_)
        ev[1].record()
        preds = v_ @ W_out.T
        g_out = (2.0 / preds.numel()) * (preds - target[:Tw]) @ W_out  # dL/dv_t for the MSE readout
        lam = torch.zeros(N, **f32)
        keep = pop.backward(v_, r_, g_out, _abi.RP_OUT_STATE0, lam, u_)
        ev[2].record()
        grad.clear()  # (the previous window's gradient goes back to the allocator first, as autograd's does)
        grad["W"] = pop.outer_product(pop.dt, u_, r_)  # what _RateWindow.backward calls: the GEMM's store form
        ev[3].record()
        torch.cuda.synchronize()
        del keep

    window(50)  # warm-up on a short window
    parts = []
    for _ in range(2):  # two full windows ("over >= 2 windows"); their mean
        window(T)
        parts.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))
    fwd, bwd, gemm = (sum(p[i] for p in parts) / len(parts) for i in range(3))
    total = fwd + bwd + gemm
    bytes_step = 4.0 * N * N
    return {
        "workload": "BPTT window T=1000, N=32768, MSE readout k=3 (BASELINE.json configs[2]); mean of two windows",
        "windows_ms": [round(sum(p), 3) for p in parts],
        "fwd_bwd_steps_per_s": T / (total * 1e-3),
        "fwd_ms_per_step": fwd / T, "bwd_ms_per_step": bwd / T, "dw_gemm_ms_per_window": gemm,
        "bwd_roofline": {"bound": "hbm", "achieved": bytes_step / (bwd / T * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": bytes_step / (bwd / T * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        "dw_gemm_roofline": {"bound": "mfma", "achieved": 2.0 * N * N * T / (gemm * 1e-3) / 1e12, "peak": 157.3,
                             "unit": "TFLOP/s", "frac": 2.0 * N * N * T / (gemm * 1e-3) / 1e12 / 157.3},
        "grad_finite": bool(torch.isfinite(grad["W"]).all().item()),
    }


def facade_fit_bptt(pop, x, N, dev, teacher_net=None):
    """The same config through the reference-compatible front end: Network.fit_bptt with trainable recurrent
    weights and a gd-trained k=3 readout, SGD, 1000-step windows (wall clock, includes torch's loss, autograd
    bookkeeping and optimizer step over the 4.3 GB weight matrix).  Teacher-student targets: the
    target is a teacher read-out of the same network's own trajectory, the student starts from a perturbed read-out;
    the SGD step size is set from the trajectory's power so that the read-out error contracts."""
    import contextlib
    from rectipy_amd import Network
    T, win = 3000, 1000
    gen = torch.Generator(device=dev).manual_seed(3)
    W_teach = torch.randn(3, N, generator=gen, device=dev) / N ** 0.5
    inp = x[:T].clone()
    with contextlib.redirect_stdout(sys.stderr):
        if teacher_net is not None:  # teacher trajectory from the forward-only network (same W, v0 = 0)
            teacher_net.reset()
            V = teacher_net.run(inp, sampling_steps=1, verbose=False, enable_grad=False).stacked("out")
            tgt = V @ W_teach.T
            power = float((V * V).sum(dim=1).mean().item())
            del V
        else:
            tgt, power = torch.randn(T, 3, generator=gen, device=dev) * 0.1, float(N)
        lr = 0.5 / (2.0 / 3.0 * power)
        net = Network(dt=pop.dt, device=str(dev))
        net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                            output_var="li_op/v", weights=pop.W, source_var="tanh_op/r", target_var="li_op/r_in",
                            train_params=["weights"], dtype=torch.float32)
        net.add_func_node("out", 3, "identity")
        net.add_edge("rnn", "out", weights=W_teach + 0.5 * torch.randn(3, N, generator=gen, device=dev) / N ** 0.5,
                     train="gd")
        net.fit_bptt(inp[:100], tgt[:100], optimizer="sgd", lr=0.0, update_steps=100, sampling_steps=100, verbose=False)
        net.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obs = net.fit_bptt(inp, tgt, optimizer="sgd", lr=lr, update_steps=win, sampling_steps=1, verbose=False)
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # sub-case of SURVEY 8d C3: only the read-out is trained (no gradient through the recurrence, so no adjoint sweep)
    with contextlib.redirect_stdout(sys.stderr):
        del net
        torch.cuda.empty_cache()
        net = Network(dt=pop.dt, device=str(dev))
        net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                            output_var="li_op/v", weights=pop.W, source_var="tanh_op/r", target_var="li_op/r_in",
                            dtype=torch.float32)
        net.add_func_node("out", 3, "identity")
        net.add_edge("rnn", "out", weights=W_teach + 0.5 * torch.randn(3, N, generator=gen, device=dev) / N ** 0.5,
                     train="gd")
        net.fit_bptt(inp[:100], tgt[:100], optimizer="sgd", lr=0.0, update_steps=100, sampling_steps=100, verbose=False)
        net.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        obs_ro = net.fit_bptt(inp, tgt, optimizer="sgd", lr=lr, update_steps=win, sampling_steps=1, verbose=False)
        torch.cuda.synchronize()
        wall_ro = time.perf_counter() - t1
        ro_losses = obs_ro["loss"]
        ro_wl = [float(ro_losses[b - 1]) for b in range(win, T + 1, win)]
        del net
    # the loss recorded at the LAST step of a window is that window's loss (reference network.py:1043-1046)
    losses = obs["loss"]
    wl = [float(losses[b - 1]) for b in range(win, T + 1, win)]
    tp = [float((tgt[b - win:b] ** 2).mean().item()) for b in range(win, T + 1, win)]
    return {"facade_fit_bptt_steps_per_s": T / wall, "facade_fit_bptt_lr": lr, "facade_fit_bptt_window_losses": wl,
            "facade_fit_readout_only_steps_per_s": T / wall_ro,
            "facade_fit_readout_only_window_relative_losses": [l / p for l, p in zip(ro_wl, [float((tgt[b - win:b] ** 2).mean().item()) for b in range(win, T + 1, win)])],
            # MSE of a window relative to the mean square of its targets (the trajectory's power grows from v0 = 0)
            "facade_fit_bptt_window_relative_losses": [l / p for l, p in zip(wl, tp)]}


def qif_extra(dev, N=65536, T=20000, cpu_threads=None):
    """BASELINE.json configs[3] AS STATED, through the front end: QIF population, N = 65536, dense
    W = rand * 20 / N, tau = k = tau_s = 1, eta = -5, v0 = -2, s0 = 0, theta = 100, v_reset = -100, dt = 1e-3, constant
    input 15, 20000 Euler steps, spike + reset path, Network.run with sampling_steps = 100 (about 49 s of GPU time)."""
    import contextlib
    from rectipy_amd import Network
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.empty(N, N, device=dev)
    for r in range(0, N, 4096):
        W[r:r + 4096] = torch.rand(4096, N, device=dev, generator=gen)
    W.mul_(20.0 / N)
    net = Network(dt=1e-3, device=str(dev))
    with contextlib.redirect_stdout(sys.stderr):
        net.add_diffeq_node("qif", "neuron_model_templates.spiking_neurons.qif.qif", input_var="qif_op/I_ext",
                            output_var="qif_op/s", weights=W, source_var="qif_op/s", target_var="qif_op/s_in",
                            spike_var="qif_op/spike", reset_var="qif_op/v", spike_threshold=100.0, spike_reset=-100.0,
                            dtype=torch.float32,
                            node_vars={"all/qif_op/eta": -5.0, "all/qif_op/tau": 1.0, "all/qif_op/k": 1.0,
                                       "all/qif_op/tau_s": 1.0, "all/qif_op/v": -2.0, "all/qif_op/s": 0.0})
    del W
    node = net.get_node("qif")
    x = torch.full((T, 1), 15.0, device=dev)
    y0 = {k_: v_.detach().clone() for k_, v_ in net.state.items()}
    net.run(x[:50], sampling_steps=50, verbose=False, enable_grad=False)  # warm-up
    net.reset(y0)  # back to v0 = -2, s0 = 0
    node.enable_spike_recording(1 << 23)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs = net.run(x, sampling_steps=100, verbose=False, enable_grad=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev = node.spikes()
    ms = wall / T * 1e3
    gbps = (4.0 * N * N + 48.0 * N) / (ms * 1e-3) / 1e9
    s_mean = obs.to_numpy("out").mean(axis=1)
    res = {"workload": f"QIF N={N} dense W = rand*20/N, eta=-5, v0=-2, s0=0, theta=100, v_reset=-100, constant input 15, "
                       f"dt=1e-3, {T} Euler steps through Network.run (sampling_steps=100), spike/reset path: "
                       f"BASELINE.json configs[3] as stated",
           "steps_per_s": T / wall, "ms_per_step": ms, "wall_s": wall, "spikes": int(ev.shape[0]),
           "first_spike_step": int(ev[:, 0].min()) if ev.shape[0] else None,
           "spikes_per_neuron_per_unit_time": ev.shape[0] / N / (T * 1e-3),
           "mean_s_first_last_sample": [float(s_mean[0]), float(s_mean[-1])], "samples": len(obs["steps"]),
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                        "note": "wall clock of the whole run (front end included) over the algorithmic bytes 4 N^2 + 48 N per step"},
           "state_finite": bool(torch.isfinite(node.current_state).all().item())}
    if cpu_threads is not None:
        from oracle.torch_eager import host_copy_spread, time_eager_qif
        Wc = host_copy_spread(node._pop.W)
        one = torch.ones(N)
        torch.set_num_threads(cpu_threads)
        warm, timed = 1, 8
        sps = time_eager_qif(Wc, one, one, torch.full((N,), -5.0), one, torch.linspace(-2.0, 99.0, N), torch.zeros(N), 15.0,
                             1e-3, 100.0, -100.0, warm, timed)
        res["cpu"] = {"value": sps, "unit": "steps/s", "cores": cpu_threads, "kind": "port",
                      "sample": f"{warm} warm-up + {timed} timed steps of the same QIF network from a spread start (threshold "
                                f"crossings in every step), torch CPU eager loop (oracle/torch_eager.py: time_eager_qif), "
                                f"fp32, no_grad"}
        del Wc
    del net, node
    torch.cuda.empty_cache()
    return res


def small_n_extra(dev, N=512, T=2000):
    """BASELINE.json configs[0] shape on the GPU (the reference's own scale: launch-bound, not HBM-bound): tanh population,
    N = 512, g = 1.5, full-width input, 2000 Euler steps through Network.run (sampling_steps = 1) and through
    Network.fit_bptt (SGD, 100-step windows, all of W trained, k = 3 read-out).  us per step, wall clock of whole calls;
    `launch_floor_us` is the duration of an empty-grid launch pair on this box (what a step cannot go below while it is
    one kernel launch per Euler step)."""
    import contextlib
    from rectipy_amd import Network
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
    x = torch.randn(T, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.5
    target = torch.randn(T, 3, device=dev, generator=gen) * 0.1
    res = {"workload": f"tanh rate RNN N={N} (BASELINE.json configs[0] shape), {T} Euler steps", "unit": "us per Euler step (wall clock)"}
    with contextlib.redirect_stdout(sys.stderr):
        net = Network(dt=1e-2, device=str(dev))
        net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32)
        net.run(x[:200], sampling_steps=1, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        res["Network_run"] = (time.perf_counter() - t0) / T * 1e6
        res["Network_run_steps_per_s"] = 1e6 / res["Network_run"]
        fit = Network(dt=1e-2, device=str(dev))
        fit.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W * 0.5, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
        fit.add_func_node("out", 3, "identity")
        fit.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="gd")
        fit.fit_bptt(x[:200], target[:200], optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
        torch.cuda.synchronize()
        res["Network_fit_bptt_sgd_100_step_windows"] = (time.perf_counter() - t0) / T * 1e6
        # RLS read-out training with an update every step (the reference's default update_steps = 1, network.py:786-856)
        rls_net = Network(dt=1e-2, device=str(dev))
        rls_net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="tanh_op/r", weights=W * 0.5, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32)
        rls_net.add_func_node("out", 3, "identity")
        rls_net.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="rls")
        rls_net.fit_rls(x[:200], target[:200], update_steps=1, sampling_steps=100, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rls_net.fit_rls(x, target, update_steps=1, sampling_steps=100, verbose=False)
        torch.cuda.synchronize()
        res["Network_fit_rls_update_every_step"] = (time.perf_counter() - t0) / T * 1e6
    # launch floor: dependent tiny kernels back to back on the stream
    z = torch.zeros(64, device=dev)
    for _ in range(200):
        z.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000):
        z.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    res["launch_floor_us"] = e0.elapsed_time(e1) / 2000 * 1e3
    res["note"] = ("r05: up to n = 1024 (1536 when n % 4 == 0; coupled circuits: members up to 1024) ALL steps of a call run in ONE launch, "
                   "forward and adjoint -- W / W^T in registers, "
                   "the vector handed from wave to wave as tagged 8-byte granules, bit-identical to a launch per step "
                   "(rectipy_amd/csrc/rp_resident.h); with one launch per step (RECTIPY_AMD_RESIDENT=0: `launch_per_step`) a step is bounded by "
                   "the dependent-launch interval of the stream and fit_bptt by the host's launch rate; fit_bptt adds the per-window work "
                   "(W^T copy, dW GEMM, read-out, loss, optimiser); fit_rls with an update every step: the RLS sweep of a window is one "
                   "launch too (P and W in registers, one exchange per update; three launches per update before)")
    # a small coupled circuit (two populations of N/2, a -> b with feedback b -> a): rp_forward_coupled runs all steps of a call in one
    # launch too (resident_coupled_kernel: the members' and the edges' rows in registers, one granule stream per member)
    from rectipy_amd import FeedbackNetwork
    h = N // 2

    def pair(train):
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        kw = lambda Wm: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=Wm, source_var="tanh_op/r", target_var="li_op/r_in",
                             dtype=torch.float32, train_params=["weights"] if train else None)
        fb.add_diffeq_node("a", tanh, **kw(gen_rows(h, 0, h, dev, g=0.8)))
        fb.add_diffeq_node("b", tanh, **kw(gen_rows(h, 0, h, dev, g=0.8)))
        fb.add_func_node("out", 3, "identity")
        fb.add_edge("a", "b", weights=gen_rows(h, 0, h, dev, g=0.5), train=train)
        fb.add_edge("b", "a", weights=gen_rows(h, 0, h, dev, g=0.5), feedback=True, train=train)
        fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev, generator=gen) / h ** 0.5, train="gd")
        return fb

    xh = x[:, :h].contiguous()

    def pair_times():
        out = {}
        with contextlib.redirect_stdout(sys.stderr):
            for key, train in (("FeedbackNetwork_run", None), ("FeedbackNetwork_fit_bptt_all_matrices", "gd")):
                fb = pair(train)
                call = (lambda: fb.run(xh, sampling_steps=1, verbose=False, enable_grad=False)) if train is None else \
                    (lambda: fb.fit_bptt(xh, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False))
                call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                call()
                torch.cuda.synchronize()
                out[key] = (time.perf_counter() - t0) / T * 1e6
                del fb
        return out
    res["two_populations_of_N_over_2"] = pair_times()
    # the same calls with one launch per step (the r04 path), for the record
    os.environ["RECTIPY_AMD_RESIDENT"] = "0"
    try:
        res["two_populations_of_N_over_2_launch_per_step"] = pair_times()
        with contextlib.redirect_stdout(sys.stderr):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
            per_step_run = (time.perf_counter() - t0) / T * 1e6
            t0 = time.perf_counter()
            fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
            torch.cuda.synchronize()
            per_step_fit = (time.perf_counter() - t0) / T * 1e6
            t0 = time.perf_counter()
            rls_net.fit_rls(x, target, update_steps=1, sampling_steps=100, verbose=False)
            torch.cuda.synchronize()
            res["launch_per_step"] = {"Network_run": per_step_run, "Network_fit_bptt_sgd_100_step_windows": per_step_fit,
                                      "Network_fit_rls_update_every_step": (time.perf_counter() - t0) / T * 1e6}
    finally:
        os.environ.pop("RECTIPY_AMD_RESIDENT", None)
    return res


def sparse_extra(dev, N=32768, p=0.1, T=1000):
    """A FIXED random coupling matrix with connection probability 0.1 (utility.random_connectivity in the reference's
    examples), streamed in the packed form that reproduces the dense arithmetic bit for bit."""
    from rectipy_amd.engine import Population
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.empty(N, N, device=dev)
    for r in range(0, N, 4096):
        blk = torch.randn(4096, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
        blk *= torch.rand(4096, N, device=dev, generator=gen) < p
        W[r:r + 4096] = blk
    del blk
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    del W
    packed = pop.pack_weights()
    x = torch.sin(torch.arange(T + 50, device=dev) * 0.0126).reshape(-1, 1).contiguous()
    pop.forward(50, x=x[:50])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pop.forward(T, x=x[50:], step0=50)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / T
    gbps = packed.stored_bytes / (ms * 1e-3) / 1e9
    res = {"workload": f"tanh rate RNN N={N}, fixed sparse W (p={p}), packed storage "
                       f"({'blocked form, ' + str(packed.block_rows) + '-row blocks' if packed.block_rows else 'per-row form'}), forward", "steps_per_s": 1e3 / ms,
           "ms_per_step": ms, "stored_bytes": packed.stored_bytes, "dense_bytes": 4 * N * N,
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS},
           "state_finite": bool(torch.isfinite(pop.v).all().item())}
    del pop, packed
    torch.cuda.empty_cache()
    res["coupled_pair"] = sparse_coupled_extra(dev, N // 2, p)
    return res


def sparse_coupled_extra(dev, h, p, T=300):
    """Two populations of N/2 neurons coupled by dense-stored edges, one a feedback edge, EVERY matrix -- inside and between
    the populations -- fixed with connection probability p, through FeedbackNetwork.run: the edge matrices are packed like
    the populations' own (rp_run.edge_sp: one packed mat-vec launch per edge, then the member's packed step; the bits of the
    dense run), next to the same network streamed densely (sparse_weights=False)."""
    import contextlib
    from rectipy_amd import FeedbackNetwork
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    gen = torch.Generator(device=dev).manual_seed(1)

    def mat(g):
        W = torch.randn(h, h, device=dev, generator=gen) * (g / (h * p) ** 0.5)
        return W * (torch.rand(h, h, device=dev, generator=gen) < p)

    mats = [mat(1.0), mat(1.0), mat(0.5), mat(0.5)]
    x = torch.randn(T, h, device=dev, generator=gen) * 0.1
    out = {"workload": f"two tanh populations of {h}, W and both edges (one feedback) fixed with p={p}, FeedbackNetwork.run",
           "unit": "us per Euler step (wall clock)"}
    with contextlib.redirect_stdout(sys.stderr):
        for mode in ("auto", False):
            fb = FeedbackNetwork(dt=1e-2, device=str(dev))
            kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32, sparse_weights=mode)
            fb.add_diffeq_node("a", tanh, **kw(mats[0].clone()))
            fb.add_diffeq_node("b", tanh, **kw(mats[1].clone()))
            fb.add_edge("a", "b", weights=mats[2].clone())
            fb.add_edge("b", "a", weights=mats[3].clone(), feedback=True)
            fb.run(x[:20], sampling_steps=20, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fb.run(x, sampling_steps=T, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
            out["packed" if mode == "auto" else "dense"] = round((time.perf_counter() - t0) / T * 1e6, 1)
            del fb
            torch.cuda.empty_cache()
    out["speedup"] = round(out["dense"] / out["packed"], 2)
    # the same pair trained: tau and eta of both populations and a read-out of width 3 (the matrices stay fixed), SGD windows of 50:
    # the adjoint streams the packed transposed matrices (rp_bwd.spt_*, rp_bwd_member.edge_spt)
    y = torch.randn(100, 3, device=dev, generator=gen) * 0.1
    with contextlib.redirect_stdout(sys.stderr):
        for mode in ("auto", False):
            fb = FeedbackNetwork(dt=1e-2, device=str(dev))
            kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                                dtype=torch.float32, train_params=["li_op/tau", "li_op/eta"], sparse_weights=mode)
            fb.add_diffeq_node("a", tanh, **kw(mats[0].clone()))
            fb.add_diffeq_node("b", tanh, **kw(mats[1].clone()))
            fb.add_func_node("out", 3, "identity")
            fb.add_edge("a", "b", weights=mats[2].clone())
            fb.add_edge("b", "a", weights=mats[3].clone(), feedback=True)
            fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev, generator=gen) / h ** 0.5, train="gd")
            fb.fit_bptt(x[:50], y[:50], optimizer="sgd", lr=1e-4, update_steps=50, sampling_steps=50, verbose=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fb.fit_bptt(x[:100], y, optimizer="sgd", lr=1e-4, update_steps=50, sampling_steps=50, verbose=False)
            torch.cuda.synchronize()
            out["fit_bptt_packed" if mode == "auto" else "fit_bptt_dense"] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
            del fb
            torch.cuda.empty_cache()
    out["fit_bptt_speedup"] = round(out["fit_bptt_dense"] / out["fit_bptt_packed"], 2)
    return out


def cpu_baseline(pop, x, dt, N):
    """Torch-eager port of the reference's per-step CPU loop on this box's host cores.  The mat-vec is DRAM-bound, so
    the default thread count (all hardware threads) is not the fastest: a short sweep picks the thread count, then 5
    warm-up + 60 timed steps at the winner (per-step cost is stationary; about 15-25 s of CPU work in total)."""
    from oracle.torch_eager import cpu_share, host_copy_spread, numa_nodes, sweep_threads, time_eager
    W = host_copy_spread(pop.W)  # pages first-touched by all hardware threads: spread over every NUMA node
    tau, k, eta = (pop.params[n].cpu() for n in ("tau", "k", "eta"))
    v = pop.v.cpu()
    warm, timed = 5, 60
    xs = x[:warm + timed].cpu().expand(warm + timed, N).contiguous()
    hw = os.cpu_count() or torch.get_num_threads()
    share = cpu_share()
    cands = sorted({t for t in (8, 16, int(round(share)), 32, 64, hw // 2, hw) if 1 <= t <= hw})
    best, _, table = sweep_threads(lambda: time_eager(W, tau, k, eta, v, xs, dt, 2, 10), cands)
    torch.set_num_threads(best)
    sps = time_eager(W, tau, k, eta, v, xs, dt, warm, timed)
    res = {"value": sps, "unit": "steps/s", "cores": best, "kind": "port",
           "sample": f"{warm} warm-up + {timed} timed Euler steps of the same N={N} network, torch {torch.__version__} "
                     f"CPU eager loop (oracle/torch_eager.py), fp32, no_grad; thread count = best of a sweep "
                     f"(2 warm-up + 10 timed steps each) over {cands} on {hw} hardware threads; W first-touched by all "
                     f"hardware threads (pages spread over the host's {numa_nodes()} NUMA node(s))",
           "hardware_threads": hw, "numa_nodes": numa_nodes(), "cpu_share_of_this_job": share,
           # the survey container (8 vCPUs) ran this loop at 17.4 steps/s = 75 GB/s: quote GPU/CPU
           # ratios against the better of the two hosts, not against a throttled share of this one
           "survey_container_probe_steps_per_s": 17.4,
           "achieved_GBps": 4.0 * N * N * sps / 1e9,  # the mat-vec's stream of W: what bounds the CPU loop
           "thread_sweep_steps_per_s": {str(t): round(v_, 2) for t, v_ in table.items()}}
    return res, best


def cpu_fwd_bwd(pop, x, dt, N, threads, T=20):
    """the CPU counterpart of the BPTT window -- the eager loop under autograd, MSE on a k=3
    read-out, backward(); window T=20 (autograd materialises a 4 GiB outer product per step)."""
    from oracle.torch_eager import time_eager_bptt
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(2)
    W_out = torch.randn(3, N, generator=gen) / N ** 0.5
    target = torch.randn(T, 3, generator=gen) * 0.1
    xs = x[:T].cpu().expand(T, N).contiguous()
    tau, k, eta = (pop.params[n].cpu() for n in ("tau", "k", "eta"))
    from oracle.torch_eager import host_copy_spread
    sps = time_eager_bptt(host_copy_spread(pop.W), tau, k, eta, pop.v.cpu(), xs, W_out, target, dt, T)
    return {"value": sps, "unit": "steps/s (fwd+bwd)", "cores": threads, "kind": "port",
            "sample": f"one autograd window of T={T} steps, N={N}, trainable W, MSE read-out k=3 "
                      f"(oracle/torch_eager.py: time_eager_bptt)"}


def batched_extra(pop, N, dev, ms_single, T=200, net=None):
    """B independent trajectories of the same network in one pass over W per step (rp_forward_batched; the "batched-input
    dense W @ state contraction" of the brief -- the reference has no batch axis and would run them one after the other).
    Up to 8 trajectories on the vector ALUs are bit-identical to unbatched runs; the matrix-core forms have their own
    summation order, mirrored bit for bit by the oracle (tests/test_batched.py)."""
    from rectipy_amd.engine import BatchedPopulation
    if N % 4:
        return {"skipped": "n % 4 != 0"}
    res = {"workload": f"tanh rate RNN N={N}, B trajectories per pass over W, {T} timed steps each, broadcast input",
           "unit": "trajectory-steps/s"}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # 2, 4 (and 8 with exact=True): vector ALUs (bit-identical to unbatched runs); 8, 16, 32: exact-fp32 matrix cores
    # (v_mfma_f32_16x16x4_f32, W and source tiles by LDS-DMA; 17..32 trajectories: role-split workgroups).  Binding roof of a step: the slower of W
    # once from HBM (4 N^2 B / 8 TB/s) and the contraction on the fp32 matrix cores (2 N^2 B flop / 157.3 TFLOP/s).
    # B8 / B16 / B32: the defaults (matrix cores from 5 trajectories); B8_exact: the vector-ALU form `exact=True` keeps
    for B, mfma in ((2, None), (4, None), (8, False), (8, None), (16, None), (32, None)):
        bp = BatchedPopulation(pop, B, mfma=mfma)
        bp.v.normal_(generator=torch.Generator(device=dev).manual_seed(B))
        bp.refresh()
        xb = torch.randn(T + 10, B, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + B))
        bp.forward(10, x=xb[:10])
        torch.cuda.synchronize()
        e0.record()
        bp.forward(T, x=xb[10:])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / T
        roof_ms = max(4.0 * N * N / (HBM_PEAK_GBPS * 1e9), 2.0 * N * N * B / 157.3e12) * 1e3
        forms = {"valu": "valu", "mfma16": "mfma_f32_16x16x4 + LDS-DMA ring", "mfma32": "mfma_f32_32x32x2"}
        res[f"B{B}" + ("_exact" if mfma is False and B == 8 else "")] = {
            "form": forms[bp.form], "trajectory_steps_per_s": B * 1e3 / ms, "ms_per_step": ms,
            "vs_unbatched": B * ms_single / ms, "W_stream_GBps": 4.0 * N * N / (ms * 1e-3) / 1e9,
            "fma_TFLOPs": 2.0 * N * N * B / (ms * 1e-3) / 1e12,
            "roof": {"bound": "hbm" if 4.0 * N * N / (HBM_PEAK_GBPS * 1e9) >= 2.0 * N * N * B / 157.3e12 else "mfma",
                     "ms_at_roof": roof_ms, "frac": roof_ms / ms},
            "state_finite": bool(torch.isfinite(bp.v).all().item())}
        del bp, xb
    torch.cuda.empty_cache()
    if net is not None:
        # the front-end path: Network.run_batch, 32 trajectories x 2000 steps, sampled every 100 steps inside the kernels.
        # Wall clock of the whole call and the peak of device memory it allocated on top of the network (r03 staged
        # [B, T, N] = 8.4 GB here; now O(B x samples x N))
        B, Tb = 32, 2000
        xb = torch.randn(B, Tb, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        net.run_batch(xb[:, :100], sampling_steps=100)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        t0 = time.perf_counter()
        obs = net.run_batch(xb, sampling_steps=100)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms = wall / Tb * 1e3
        roof_ms = max(4.0 * N * N / (HBM_PEAK_GBPS * 1e9), 2.0 * N * N * B / 157.3e12) * 1e3
        res["Network_run_batch_B32"] = {
            "workload": f"Network.run_batch: {B} trajectories x {Tb} steps, sampling_steps=100, broadcast input",
            "wall_s": wall, "ms_per_step": ms, "trajectory_steps_per_s": B * Tb / wall, "roof_frac": roof_ms / ms,
            "samples_per_trajectory": len(obs[0]["steps"]),
            "peak_device_memory_above_the_network_MB": (torch.cuda.max_memory_allocated(dev) - base) / 1e6,
            "staging_all_steps_would_take_MB": 4.0 * B * Tb * N / 1e6}
        del xb, obs
        torch.cuda.empty_cache()
    return res


def rls_extra(dev, n_in=32768, k=3, T=64, cpu_threads=None):
    """recursive-least-squares read-out training (reference RLS.update, rectipy/edges.py:227-234) at
    n_in = 32768: P is 4.3 GB.  The reference makes three passes over P per update (12 n_in^2 bytes + an n_in^2 temporary).
    Here (the default with the reference's default beta = 1, chosen explicitly for this leg's beta = 0.99): the BLOCKED symmetric sweep -- the inputs of a window are known in advance, so
    one read-modify-write pass over the blocks of P on and above the diagonal applies 8 updates and forms the products with
    the next 8 inputs (rp_rls.block; the z vectors inside a block follow from a double-precision recurrence on n_in-vectors).
    Reported next to it: the symmetric pass per update (r03 / r04 first half) and the dense pass per update (r02)."""
    from rectipy_amd.rls import RLS
    gen = torch.Generator(device=dev).manual_seed(5)
    X = torch.randn(T, n_in, device=dev, generator=gen) * 0.1
    Y = torch.randn(T, k, device=dev, generator=gen)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms, finite = {}, True
    for form in ("blocked", "symmetric", "dense"):
        edge = RLS(n_in, k, beta=0.99, alpha=1.0).to(dev)
        edge.symmetric = form != "dense"
        edge.block = 8 if form == "blocked" else 1   # (8 is the default with beta = 1, the reference's default; here beta = 0.99)
        Tf = T if form == "blocked" else 20
        edge.sweep(X[:16], Y[:16], 1)
        edge.sweep(X[:Tf], Y[:Tf], 1)   # (warm-up: a call of the timed length -- the first pass pattern over the 4 GiB of P after
        #                                  its allocation runs 15 % slower than every later one, scripts/probes/rls_ab_libs.py)
        torch.cuda.synchronize()
        e0.record()
        edge.sweep(X[:Tf], Y[:Tf], 1)
        e1.record()
        torch.cuda.synchronize()
        # blocked: everything a call costs per update; pass per update: Tf fused passes + the initial mat-vec pass
        ms[form] = e0.elapsed_time(e1) / (Tf if form == "blocked" else Tf + 1)
        finite = finite and bool(torch.isfinite(edge.P).all().item())
        block = edge.BLOCK
        del edge
        torch.cuda.empty_cache()
    nb = (n_in + 1023) // 1024
    dense_bytes = 8.0 * n_in * n_in
    sym_bytes = 8.0 * 1024.0 * 1024.0 * nb * (nb + 1) / 2 if n_in % 1024 == 0 else dense_bytes * (nb + 1) / (2.0 * nb)
    gb = {"blocked": sym_bytes / block / (ms["blocked"] * 1e-3) / 1e9, "symmetric": sym_bytes / (ms["symmetric"] * 1e-3) / 1e9,
          "dense": dense_bytes / (ms["dense"] * 1e-3) / 1e9}
    roof = lambda form, bytes_, note: {"bound": "hbm", "achieved": gb[form], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": gb[form] / HBM_PEAK_GBPS, "algorithmic_bytes_per_update": bytes_, "note": note}
    res = {"workload": f"RLS read-out training, n_in={n_in}, n_out={k}, {T} consecutive updates (rp_rls_sweep)",
           "updates_per_s": 1e3 / ms["blocked"], "ms_per_update": ms["blocked"],
           "form": f"blocked symmetric sweep: one read-modify-write pass over the 1024 x 1024 blocks of P on and above the diagonal "
                   f"per {block} updates (their rank-1 updates + the products with the next {block} inputs); the block's scalars "
                   "(k, errors, the z vectors' coefficients) from its Gram matrix in double precision: two small launches per BLOCK "
                   "(r05; r04: two per update); bit-exact against the oracle, as close to the reference's float64 run as a pass "
                   "per update (G31 / G34)",
           "roofline": roof("blocked", sym_bytes / block,
                            f"the pass's bytes (blocks on and above the diagonal, read + written once) / {block} updates; a call of U updates "
                            f"makes U / {block} + 1 passes (the first read-only, the last update-only); the Y-from-partials kernel, the "
                            "Gram / apply launches and the column-partial traffic are the rest of the time"),
           "symmetric_pass_per_update": {"ms_per_update": ms["symmetric"], "updates_per_s": 1e3 / ms["symmetric"],
                                         "roofline": roof("symmetric", sym_bytes, "r04 first half: 0.76 ms")},
           "dense_pass": {"ms_per_update": ms["dense"], "updates_per_s": 1e3 / ms["dense"],
                          "roofline": roof("dense", dense_bytes, "an in-place read-modify-write of 4 GiB reaches 6.7-6.8 TB/s on this chip")},
           "speedup_over_dense_pass": ms["dense"] / ms["blocked"],
           "speedup_over_symmetric_pass_per_update": ms["symmetric"] / ms["blocked"],
           "state_finite": finite}
    del X, Y
    torch.cuda.empty_cache()
    if cpu_threads is not None:
        from oracle.torch_eager import time_eager_rls
        torch.set_num_threads(cpu_threads)
        ups = time_eager_rls(n_in, k, 3)
        res["cpu"] = {"value": ups, "unit": "updates/s", "cores": cpu_threads, "kind": "port",
                      "sample": "1 warm-up + 3 timed updates of the reference's RLS.update, op for op, torch CPU eager "
                                "(oracle/torch_eager.py: time_eager_rls), fp32"}
    return res


def stepwise_extra(dev):
    """What the paths cost that advance ONE Euler step per call (us per step, wall clock incl. the Python graph walk), at
    N = 4096 and N = 32768: (i) a user loop over net.forward(x) (reference rectipy_tests/test_network.py:327-333);
    (ii) inp -> LinearMemoryFilter -> rnn through Network.run -- r03: the edge is evaluated per window, so the population
    takes the fused-window plan (r05: the edge's memory is one single-workgroup launch per window, rp_edge_window); the same
    graph forced onto the step-by-step walk is what r02 paid; (iii) a
    FeedbackNetwork of two populations of N/2 neurons (a -> b, feedback b -> a; dense N/2 x N/2 edges) -- r04: one
    rp_forward_coupled call per window, both edge matrices streamed by the targets' step kernels (r03: a Python walk per
    step with the edge products in rocBLAS); the same graph forced onto the per-step walk is reported next to it; (iv) the
    same two populations trained: FeedbackNetwork.fit_bptt, both W and both edge matrices trainable plus a read-out of
    width 3, SGD windows of 50 steps -- r04: one rp_forward_coupled and one rp_backward_coupled call per window (+ the
    weight-gradient GEMMs and the optimiser); at N = 4096 also with the adjoint forced onto autograd over the per-step walk."""
    import contextlib
    from rectipy_amd import FeedbackNetwork, Network
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    res = {"unit": "us per Euler step (wall clock)", "sizes": {}}

    def node_kw(W):
        return dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                    target_var="li_op/r_in", dtype=torch.float32)

    def timed(fn, steps):
        fn(max(2, steps // 10))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    with contextlib.redirect_stdout(sys.stderr):
        for N, T in ((4096, 1000), (32768, 200)):
            m = 8
            gen = torch.Generator(device=dev).manual_seed(N)
            W = gen_rows(N, 0, N, dev)
            W_in = torch.randn(N, m, device=dev, generator=gen)
            x_full = torch.randn(T, N, device=dev, generator=gen) * 0.1
            x_m = torch.randn(T, m, device=dev, generator=gen)
            rec = {}
            # (i) user loop over net.forward
            net = Network(dt=1e-2, device=str(dev))
            net.add_diffeq_node("rnn", tanh, **node_kw(W))
            net.compile()

            def user_loop(steps):
                with torch.no_grad():
                    for t in range(steps):
                        net.forward(x_full[t])
            rec["user_loop_net_forward"] = timed(user_loop, T)
            rec["fused_Network_run_same_graph"] = timed(
                lambda steps: net.run(x_full[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
            del net
            # (ii) a delay + filter edge in front of the population
            for forced in (False, True):
                net = Network(dt=1e-2, device=str(dev))
                net.add_func_node("inp", m, "identity")
                net.add_diffeq_node("rnn", tanh, **node_kw(W))
                net.add_edge("inp", "rnn", weights=W_in, delays=torch.arange(m) % 3,
                             filter_weights=torch.eye(m) * 0.5)
                net._force_stepwise = forced
                key = "memory_filter_edge_" + ("step_by_step_walk" if forced else "windowed_edge_fused_population")
                rec[key] = timed(lambda steps: net.run(x_m[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
                del net
            del W, W_in, x_full
            torch.cuda.empty_cache()
            # (iii) two populations with a feedback edge
            h = N // 2
            fb = FeedbackNetwork(dt=1e-2, device=str(dev))
            fb.add_diffeq_node("a", tanh, **node_kw(gen_rows(h, 0, h, dev)))
            fb.add_diffeq_node("b", tanh, **node_kw(gen_rows(h, 0, h, dev)))
            fb.add_edge("a", "b", weights=gen_rows(h, 0, h, dev, g=0.5))
            fb.add_edge("b", "a", weights=gen_rows(h, 0, h, dev, g=0.5), feedback=True)
            x_h = torch.randn(T, h, device=dev, generator=gen) * 0.1
            rec["feedback_two_populations_of_N_over_2"] = timed(
                lambda steps: fb.run(x_h[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
            fb._force_stepwise = True
            rec["feedback_two_populations_step_by_step_walk"] = timed(
                lambda steps: fb.run(x_h[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
            del fb
            torch.cuda.empty_cache()
            # (iii-b) r05: the same pair with memory / an activation node on the feed-forward path, inside the library call:
            #   a -(LinearFilter: y <- F y + x, F as large as a recurrent matrix)-> b: five matrices of (N/2)^2 per step
            #   a -> tanh node -> b: the activation node is a member of the call, its two edge matrices stream like the rest
            for key, build in (
                ("feedback_pair_with_LinearFilter_edge",
                 lambda f: f.add_edge("a", "b", weights=gen_rows(h, 0, h, dev, g=0.5), filter_weights=gen_rows(h, 0, h, dev, g=0.5))),
                ("feedback_pair_with_tanh_node_between",
                 lambda f: (f.add_func_node("mid", h, "tanh"), f.add_edge("a", "mid", weights=gen_rows(h, 0, h, dev, g=0.5)),
                            f.add_edge("mid", "b", weights=gen_rows(h, 0, h, dev, g=0.5))))):
                fb = FeedbackNetwork(dt=1e-2, device=str(dev))
                fb.add_diffeq_node("a", tanh, **node_kw(gen_rows(h, 0, h, dev)))
                fb.add_diffeq_node("b", tanh, **node_kw(gen_rows(h, 0, h, dev)))
                build(fb)
                fb.add_edge("b", "a", weights=gen_rows(h, 0, h, dev, g=0.5), feedback=True)
                rec[key] = timed(lambda steps: fb.run(x_h[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
                rec[key + "_per_matrix"] = rec[key] / 5
                if N <= 4096:
                    fb._force_stepwise = True
                    rec[key + "_step_by_step_walk"] = timed(
                        lambda steps: fb.run(x_h[:steps], sampling_steps=steps, verbose=False, enable_grad=False), T)
                del fb
                torch.cuda.empty_cache()
            rec["feedback_two_populations_per_matrix"] = rec["feedback_two_populations_of_N_over_2"] / 4
            # (iv) the two populations trained through the cycle
            train_kw = lambda Wm: dict(node_kw(Wm), train_params=["weights"])
            y_h = torch.randn(T, 3, device=dev, generator=gen) * 0.1
            for walk in ((False, True) if N <= 4096 else (False,)):
                fb = FeedbackNetwork(dt=1e-2, device=str(dev))
                fb.add_diffeq_node("a", tanh, **train_kw(gen_rows(h, 0, h, dev, g=0.8)))
                fb.add_diffeq_node("b", tanh, **train_kw(gen_rows(h, 0, h, dev, g=0.8)))
                fb.add_func_node("out", 3, "identity")
                fb.add_edge("a", "b", weights=gen_rows(h, 0, h, dev, g=0.5), train="gd")
                fb.add_edge("b", "a", weights=gen_rows(h, 0, h, dev, g=0.5), feedback=True, train="gd")
                fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev, generator=gen) / h ** 0.5, train="gd")
                fb._force_stepwise_adjoint = walk
                Tf = T // 50 * 50 if not walk else 200

                def fit(steps, fb=fb):
                    steps = max(50, steps // 50 * 50)
                    fb.fit_bptt(x_h[:steps], y_h[:steps], optimizer="sgd", lr=1e-4, update_steps=50, sampling_steps=50, verbose=False)
                fit(50)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fit(Tf)
                torch.cuda.synchronize()
                rec["feedback_fit_bptt" + ("_adjoint_on_the_per_step_walk" if walk else "")] = (time.perf_counter() - t0) / Tf * 1e6
                if not walk:
                    # plain SGD: the update of a trained matrix happens in the gradient GEMM in every window but the call's last
                    # complete one (Network._fused_sgd_plan), so a longer fit shows more of it; and the optimiser's passes instead
                    x3, y3 = x_h[:Tf].repeat(3, 1), y_h[:Tf].repeat(3, 1)
                    for key_, env_ in (("feedback_fit_bptt_three_times_the_windows", None), ("feedback_fit_bptt_optimizer_passes", "0")):
                        if env_ is not None:
                            os.environ["RECTIPY_AMD_FUSED_SGD"] = env_
                        try:
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            fb.fit_bptt(x3, y3, optimizer="sgd", lr=1e-4, update_steps=50, sampling_steps=50, verbose=False)
                            torch.cuda.synchronize()
                            rec[key_] = (time.perf_counter() - t0) / (3 * Tf) * 1e6
                        finally:
                            os.environ.pop("RECTIPY_AMD_FUSED_SGD", None)
                    del x3, y3
                del fb
                torch.cuda.empty_cache()
            # (iv-b) r05 (ABI v18): the same training with an activation node / an edge with memory inside the cycle -- the gradient
            # passes through them inside rp_backward_coupled (five matrices of (N/2)^2 forward, five transposed ones backward)
            for key, build in (
                ("feedback_fit_bptt_through_tanh_node",
                 lambda f: (f.add_func_node("mid", h, "tanh"), f.add_edge("a", "mid", weights=gen_rows(h, 0, h, dev, g=0.5), train="gd"),
                            f.add_edge("mid", "b", weights=gen_rows(h, 0, h, dev, g=0.5), train="gd"))),
                ("feedback_fit_bptt_through_LinearFilter_edge",
                 lambda f: f.add_edge("a", "b", weights=gen_rows(h, 0, h, dev, g=0.5), filter_weights=gen_rows(h, 0, h, dev, g=0.5),
                                      train="gd"))):
                for walk in ((False, True) if N <= 4096 else (False,)):
                    fb = FeedbackNetwork(dt=1e-2, device=str(dev))
                    fb.add_diffeq_node("a", tanh, **train_kw(gen_rows(h, 0, h, dev, g=0.8)))
                    fb.add_diffeq_node("b", tanh, **train_kw(gen_rows(h, 0, h, dev, g=0.8)))
                    fb.add_func_node("out", 3, "identity")
                    build(fb)
                    fb.add_edge("b", "a", weights=gen_rows(h, 0, h, dev, g=0.5), feedback=True, train="gd")
                    fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev, generator=gen) / h ** 0.5, train="gd")
                    fb._force_stepwise_adjoint = walk
                    Tf = T // 50 * 50 if not walk else 100

                    def fit(steps, fb=fb):
                        steps = max(50, steps // 50 * 50)
                        fb.fit_bptt(x_h[:steps], y_h[:steps], optimizer="sgd", lr=1e-4, update_steps=50, sampling_steps=50, verbose=False)
                    fit(50)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fit(Tf)
                    torch.cuda.synchronize()
                    rec[key + ("_adjoint_on_the_per_step_walk" if walk else "")] = (time.perf_counter() - t0) / Tf * 1e6
                    del fb
                    torch.cuda.empty_cache()
            del x_h, y_h
            torch.cuda.empty_cache()
            res["sizes"][f"N{N}"] = {k_: round(v_, 2) for k_, v_ in rec.items()}
    return res


def one_rank_exchange_extra(timeout=420):
    """The row-sharded path with ONE rank, once per exchange mode, each in a fresh CHILD process of this benchmark (the
    parent keeps running; nothing re-executes a process that has touched the GPU): `bench.py --gpus 1 --force-dist
    --exchange rccl|direct`, forward (200 timed steps of the N = 32768 network) and a forward + adjoint window.  With one
    rank no data moves, so this is the in-library cost of the exchange machinery per step -- RCCL's all-gather /
    reduce-scatter launches, the direct exchange's push + stream signal -- in a record the driver keeps; the scaling curve
    itself needs the multi-GPU node."""
    import subprocess
    res = {"note": "one rank: the exchange moves no data; per-step cost of the in-library exchange path, child processes of bench.py"}
    for mode in ("rccl", "direct"):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-dist", "--exchange", mode, "--steps", "200",
               "--warmup", "20", "--no-cpu-baseline", "--no-extra", "--config5-neurons", "0"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            fb = (line.get("extra") or {}).get("sharded_fwd_bwd") or {}
            res[mode] = {"forward_steps_per_s": line["value"], "forward_ms_per_step": line["ms_per_step"],
                         "exchange": line["config"]["exchange"], "kernel_plus_exchange_ms": line["roofline"]["avg_launch_ms"],
                         "fwd_bwd": {k: fb[k] for k in ("fwd_bwd_steps_per_s", "ms_per_step_pair", "cross_rank_sum", "grad_finite",
                                                        "skipped", "error") if k in fb}}
        except Exception as e:  # reported, never fatal
            res[mode] = {"error": f"{type(e).__name__}: {e}"}
    return res


def ridge_extra(net, x, N, dev, T=2000, sampling=10, k=3):
    """Network.fit_ridge -- state collection over T steps on the fused window (sampled every
    `sampling` steps) + the closed-form read-out solve (float64 Gram matrix and solve in torch, dual form S x S).  The
    phases are timed separately as well (each followed by a synchronize): what the call spends outside the step kernels."""
    import contextlib
    from rectipy_amd import Network
    gen = torch.Generator(device=dev).manual_seed(4)
    targets = torch.randn(T, k, generator=gen, device=dev)
    with contextlib.redirect_stdout(sys.stderr):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obs = net.fit_ridge(x[:T], targets, sampling_steps=sampling, alpha=1e-2, verbose=False, add_readout_node=False)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        # the same phases one by one
        net.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_obs = net.run(x[:T], sampling_steps=sampling, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        X = run_obs.stacked("out")
        Y = targets[run_obs["steps"]]
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        w2 = Network._ridge_solve(X, Y, 1e-2)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        y = X @ w2
        torch.cuda.synchronize()
        t5 = time.perf_counter()
    w = obs["w_out"]
    return {"workload": f"Network.fit_ridge, N={N}, {T} steps, sampling_steps={sampling}, k={k}",
            "steps_per_s": T / wall, "wall_s": wall, "w_out_shape": list(w.shape),
            "phases_ms": {"state_collection_Network_run": (t2 - t1) * 1e3, "stack_samples_and_targets": (t3 - t2) * 1e3,
                          "float64_gram_and_solve": (t4 - t3) * 1e3, "predictions": (t5 - t4) * 1e3,
                          "step_kernels_alone_at_the_measured_launch_time": None,
                          # wall of the first call minus the phases of the second: one-time library set-up (the first
                          # torch.linalg.solve of a process creates its rocSOLVER / hipBLAS handles)
                          "one_time_setup_in_the_first_call": wall * 1e3 - (t5 - t1) * 1e3},
            "w_out_finite": bool(torch.isfinite(w).all().item()) and bool(torch.isfinite(y).all().item())}


def sharded_fwd_bwd_leg(sp, x, world, rank, dev, T=50):
    """fwd+bwd of the row-sharded network (BASELINE configs[2] shape per shard) through the exchange in use: forward window
    with the per-step all-gather, adjoint sweep with the per-step cross-rank sum of W_p^T u_p (RCCL reduce-scatter, or the
    direct exchange's rank-ordered sum), dW_p = dt U_p^T R as one rectangular MFMA GEMM per rank.  Collective.  Runs only
    when --exchange was given explicitly (the driver's default command line does not pay for it)."""
    import torch.distributed as dist
    from rectipy_amd import _abi
    pop = sp.pop
    rows, n = pop.rows, pop.n
    z = lambda *shape: torch.zeros(*shape, device=dev)
    saved_v, saved_r, full = z(T, rows), z(T, rows), z(T, n)
    g_out = torch.randn(T, rows, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + rank)) * 1e-3
    lam, u = z(rows), z(T, rows)

    def window():
        lam.zero_()
        sp.pop.forward(T, x=x[:T], save_v=saved_v, save_src=saved_r, save_src_full=full, comm=sp.comm) if sp.comm is not None \
            else sp.forward(T, x=x[:T])
        keep = sp.backward(saved_v, saved_r, g_out, _abi.RP_OUT_STATE0, lam, u)
        gW = torch.empty(rows, n, device=dev)
        pop.backend.outer_store_rect(rows, n, T, float(pop.dt), u.data_ptr(), rows, full.data_ptr(), n, gW.data_ptr(), n, dev)
        return keep, gW

    if sp.comm is None:
        return None if rank else {"skipped": "no in-library exchange on this run (host loop): not timed"}
    window()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep, gW = window()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    wall = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    finite = torch.tensor([int(torch.isfinite(lam).all().item() and torch.isfinite(gW).all().item())], device=dev)
    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    del keep, gW
    if rank:
        return None
    wall = float(wall.item())
    return {"workload": f"fwd+bwd window of {T} steps, N={n} row-sharded x{world}, incl. the shard's dW GEMM",
            "fwd_bwd_steps_per_s": T / wall, "ms_per_step_pair": wall / T * 1e3,
            "cross_rank_sum": "direct exchange: partials added in rank order (deterministic)" if sp.direct is not None
            else "RCCL reduce-scatter enqueued in-library", "grad_finite": bool(finite.item())}


def config5_leg(n5, steps, world, rank, dev, dt):
    """BASELINE.json configs[4]: tanh rate network with N = 262144 neurons row-sharded over the ranks (W generated
    shard-wise on device, 34.4 GB per GPU at 8 ranks), one RCCL all-gather of the source vector per step.  Collective:
    every rank runs it.  Returns the record on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    from rectipy_amd.sharded import ShardedPopulation, shard_bounds
    row0, rows = shard_bounds(n5, world, rank)
    if 4.0 * rows * n5 > 200e9:
        return None if rank else {"skipped": f"a {rows} x {n5} shard does not fit one GPU"}
    sp = ShardedPopulation("li_tanh", gen_rows(n5, row0, rows, dev), dict(tau=10.0, k=1.0, eta=0.0), dt, dev, n=n5)
    exchange = "torch.distributed all_gather_into_tensor per step (host loop)"
    if sp.check_inlib_exchange(2):  # collective: every rank calls it
        exchange = "RCCL all-gather enqueued in-library"
    warm = 20
    tt = torch.arange(warm + steps, device=dev, dtype=torch.float32) * dt
    x = torch.sin(2 * torch.pi * 0.2 * tt).reshape(-1, 1).contiguous()
    sp.forward(warm, x=x[:warm])
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    sp.forward(steps, x=x[warm:], step0=warm)
    e1.record()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    wall = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    finite = torch.tensor([int(torch.isfinite(sp.pop.v).all().item())], device=dev)
    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    ms_dev = e0.elapsed_time(e1) / steps
    del sp
    torch.cuda.empty_cache()
    if rank:
        return None
    wall = float(wall.item())
    algo = 4.0 * rows * n5 + 4.0 * (7 * rows + n5)
    gbps = algo / (ms_dev * 1e-3) / 1e9
    return {"workload": f"tanh rate RNN N={n5}, dense W row-sharded x{world} ({rows} rows = {4.0 * rows * n5 / 1e9:.1f} GB "
                        f"per GPU), {steps} timed steps after {warm} warm-up; {exchange} (BASELINE.json configs[4])",
            "steps_per_s": steps / wall, "ms_per_step": wall / steps * 1e3, "n_gpus": world,
            "per_gpu_roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": gbps / HBM_PEAK_GBPS, "rank0_ms_per_step_incl_exchange": ms_dev,
                                 "algorithmic_bytes_per_step_per_gpu": algo},
            "state_finite": bool(finite.item())}


if __name__ == "__main__":
    main()
