"""Row-sharded integration, world_size 2 (and an 8-rank rehearsal of the 8-GPU node), gloo on CPU (oracle
arithmetic): the sharded FORWARD trajectory equals the single-process one bit for bit (`array_equal`), because the
per-row summation tree does not depend on the sharding.  Sharded GRADIENTS are compared with a stated tolerance
(rtol 1e-5 / atol 1e-6 on weights after SGD windows): the cross-rank sum of the adjoint's full-length partials is the
collective's (gloo all-reduce here, RCCL reduce-scatter on the GPU box), whose reduction order is not the
single-process chunk order, so they agree to fp32 rounding, not bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N, T, WORLD = 256, 30, 2


def _data(N=N):
    g = torch.Generator().manual_seed(0)
    W = torch.randn(N, N, generator=g) * 1.5 / N ** 0.5
    params = dict(tau=torch.rand(N, generator=g) * 5 + 5, k=torch.rand(N, generator=g) + 0.5,
                  eta=torch.randn(N, generator=g) * 0.1)
    v0 = torch.randn(N, generator=g)
    x = torch.randn(T, N, generator=g) * 0.5
    return W, params, v0, x


def _worker(rank, port, model, out_dir, WORLD=WORLD, N=N):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd import _abi
        from rectipy_amd.sharded import ShardedPopulation, shard_bounds
        W, params, v0, x = _data(N)
        row0, rows = shard_bounds(N, WORLD, rank)
        sl = slice(row0, row0 + rows)
        kw = {}
        if model == "qif":
            params = dict(tau=1.0, k=1.0, eta=-5.0 + params["eta"][sl], tau_s=1.0)
            W = W.abs() * 2
            kw = dict(s0_local=torch.zeros(rows), theta=5.0, v_reset=-5.0)
            x = x + 15.0
        else:
            params = {k: v[sl] for k, v in params.items()}
        sp = ShardedPopulation("li_tanh" if model == "li" else "qif", W[sl], params, 1e-2, "cpu", n=N,
                               v0_local=v0[sl], backend=OracleBackend(), **kw)
        traj = torch.zeros(T, rows)
        samples = torch.zeros(T, rows)
        n_s = sp.forward(T, x=x[:, sl], traj=traj, out_var=_abi.RP_OUT_SOURCE, cutoff=3, sampling_steps=4,
                         samples=samples)
        np.savez(os.path.join(out_dir, f"{model}_{rank}.npz"), traj=traj.numpy(), samples=samples[:n_s].numpy(),
                 v=sp.pop.v.numpy(), src=sp.pop.source.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,WORLD,N", [("li", 2, 256), ("qif", 2, 256),
                                           # rehearsal of the 8-GPU node: 128-row shards, and the smallest legal shard
                                           # (8 rows: shard offsets must be multiples of the 8-row group of the walk)
                                           ("li", 8, 1024), ("qif", 8, 1024), ("li", 8, 64), ("qif", 8, 64)])
def test_rank_shards_reproduce_single_process(tmp_path, oracle_backend, model, WORLD, N):
    from rectipy_amd import _abi
    from rectipy_amd.engine import Population
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, model, str(tmp_path), WORLD, N), nprocs=WORLD, join=True)

    W, params, v0, x = _data(N)
    kw = {}
    if model == "qif":
        params = dict(tau=1.0, k=1.0, eta=-5.0 + params["eta"], tau_s=1.0)
        W = W.abs() * 2
        kw = dict(s0=torch.zeros(N), theta=5.0, v_reset=-5.0)
        x = x + 15.0
    full = Population("li_tanh" if model == "li" else "qif", W, params, 1e-2, "cpu", v0=v0, backend=oracle_backend, **kw)
    traj = torch.zeros(T, N)
    samples = torch.zeros(T, N)
    n_s = full.forward(T, x=x, traj=traj, out_var=_abi.RP_OUT_SOURCE, cutoff=3, sampling_steps=4, samples=samples)
    parts = [np.load(tmp_path / f"{model}_{r}.npz") for r in range(WORLD)]
    assert np.array_equal(np.concatenate([p["traj"] for p in parts], axis=1), traj.numpy())
    assert np.array_equal(np.concatenate([p["samples"] for p in parts], axis=1), samples[:n_s].numpy())
    assert np.array_equal(np.concatenate([p["v"] for p in parts]), full.v.numpy())
    for p in parts:  # every rank holds the full, identical source vector after the last all-gather
        assert np.array_equal(p["src"], full.source.numpy())
    if model == "qif":
        assert np.abs(np.diff(traj.numpy(), axis=0)).max() > 0.5, "no spikes in the sharded QIF test"


def _net_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork
        W, params, v0, x, W_in, W_out, u = _net_data()
        net = ShardedNetwork(1e-2, device="cpu", backend=OracleBackend())
        net.add_func_node("inp", 3, "identity")
        net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                            input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                            node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                       "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
        net.add_func_node("out", 2, "tanh")
        net.add_edge("inp", "rnn", weights=W_in.numpy())
        net.add_edge("rnn", "out", weights=W_out.numpy())
        obs = net.run(u.numpy(), sampling_steps=3, cutoff=2, verbose=False, enable_grad=False,
                      record_vars=[("rnn", "li_op/v", False)])
        np.savez(os.path.join(out_dir, f"net_{rank}.npz"), out=obs.to_numpy("out"), v=obs.to_numpy(("rnn", "li_op/v")),
                 state=net.state["rnn"].numpy(), steps=np.asarray(obs["steps"]))
    finally:
        dist.destroy_process_group()


TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"


def _net_data():
    W, params, v0, x = _data()
    g = torch.Generator().manual_seed(3)
    return W, params, v0, x, torch.randn(N, 3, generator=g), torch.randn(2, N, generator=g) / N ** 0.5, \
        torch.randn(T, 3, generator=g)


def test_sharded_network_front_end(tmp_path, oracle_backend):
    """ShardedNetwork.run (inp -> rnn -> tanh read-out) on 2 gloo ranks == Network.run in one process."""
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_net_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    W, params, v0, x, W_in, W_out, u = _net_data()
    net = Network(1e-2, device="cpu", backend=oracle_backend)
    net.add_func_node("inp", 3, "identity")
    net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                        node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                   "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
    net.add_func_node("out", 2, "tanh")
    net.add_edge("inp", "rnn", weights=W_in.numpy())
    net.add_edge("rnn", "out", weights=W_out.numpy())
    obs = net.run(u.numpy(), sampling_steps=3, cutoff=2, verbose=False, enable_grad=False,
                  record_vars=[("rnn", "li_op/v", False)])
    parts = [np.load(tmp_path / f"net_{r}.npz") for r in range(WORLD)]
    for p in parts:
        assert list(p["steps"]) == list(obs["steps"])
        assert np.abs(p["out"] - obs.to_numpy("out")).max() < 1e-6      # read-out: partial sums all-reduced
    assert np.array_equal(np.concatenate([p["v"] for p in parts], axis=1), obs.to_numpy(("rnn", "li_op/v")))
    assert np.array_equal(np.concatenate([p["state"] for p in parts]), net.state["rnn"].numpy())


def _write_build(cls, where, train):
    W, params, v0, x = _data()
    net = cls(1e-2, **where)
    net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32, train_params=train,
                        node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                   "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
    return net, x[:, :1].numpy().copy(), (v0 * 0.5 + 0.25)


def _write_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork, shard_bounds
        row0, rows = shard_bounds(N, WORLD, rank)
        for train in (None, ["weights"]):
            net, x, v = _write_build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()), train)
            net.run(x[:5], verbose=False)
            net.state["rnn"][:] = v[row0:row0 + rows]     # every rank writes its slice of the state
            obs = net.run(x[5:12], verbose=False)
            np.save(os.path.join(out_dir, f"write_{'t' if train else 'n'}_{rank}.npy"), obs.to_numpy("out"))
    finally:
        dist.destroy_process_group()


def test_state_write_on_a_sharded_population_reaches_every_rank(tmp_path, oracle_backend):
    """`net.state[...][:] = v` between two runs of a ShardedNetwork: the rank's slice of the source vector r = tanh(v)
    changes with the write and the other ranks need it before the next step (one extra all-gather)."""
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_write_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for train, tag in ((None, "n"), (["weights"], "t")):
        net, x, v = _write_build(Network, dict(device="cpu", backend=oracle_backend), train)
        net.run(x[:5], verbose=False)
        net.state["rnn"][:] = v
        want = net.run(x[5:12], verbose=False).to_numpy("out")
        got = np.concatenate([np.load(tmp_path / f"write_{tag}_{r}.npy") for r in range(WORLD)], axis=1)
        assert np.array_equal(got[0], v.numpy())
        assert np.array_equal(got, want)


def _bptt_worker(rank, port, out_dir, WORLD=WORLD):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork
        net, node, edge, x, tgt = _bptt_build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()))
        obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=15, sampling_steps=1, verbose=False)
        np.savez(os.path.join(out_dir, f"bptt_{rank}.npz"), W=node["weights"].detach().numpy(),
                 tau=node["li_op/tau"].detach().numpy(), k=node["li_op/k"].detach().numpy(),
                 eta=node["li_op/eta"].detach().numpy(), W_out=edge.weights.detach().numpy(),
                 loss=obs.to_numpy("loss"), out=obs.to_numpy("out"))
    finally:
        dist.destroy_process_group()


def _bptt_build(cls, where):
    W, params, v0, x = _data()
    g = torch.Generator().manual_seed(9)
    W_out = (torch.randn(2, N, generator=g) / N ** 0.5).numpy()
    tgt = (torch.randn(T, 2, generator=g) * 0.1).numpy()
    net = cls(1e-2, **where)
    node = net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                               input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                               train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"],
                               node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                          "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
    net.add_func_node("out", 2, "identity")
    edge = net.add_edge("rnn", "out", weights=W_out, train="gd")
    return net, node, edge, x[:, :1].numpy().copy(), tgt


@pytest.mark.parametrize("WORLD", [2, 8])
def test_sharded_fit_bptt_matches_single_process(tmp_path, oracle_backend, WORLD):
    """Sharded BPTT (reduce-scatter of W^T u per step, local dW GEMM) on 2 and on 8 gloo ranks reproduces the
    single-process fit_bptt: two SGD windows, recurrent weights + per-neuron parameters + read-out.  Tolerance: the
    cross-rank sum is the collective's, so parameters after the windows agree to 1e-6 absolute (2e-6 for the per-neuron
    vectors), not bit for bit -- unlike the forward pass above."""
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bptt_worker, args=(port, str(tmp_path), WORLD), nprocs=WORLD, join=True)
    net, node, edge, x, tgt = _bptt_build(Network, dict(device="cpu", backend=oracle_backend))
    obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=15, sampling_steps=1, verbose=False)
    parts = [np.load(tmp_path / f"bptt_{r}.npz") for r in range(WORLD)]
    W_sh = np.concatenate([p["W"] for p in parts], axis=0)
    assert np.abs(W_sh - node["weights"].detach().numpy()).max() < 1e-6
    for key, name in (("tau", "li_op/tau"), ("k", "li_op/k"), ("eta", "li_op/eta")):
        got = np.concatenate([p[key] for p in parts])
        assert np.abs(got - node[name].detach().numpy()).max() < 2e-6, key
    W_out_sh = np.concatenate([p["W_out"] for p in parts], axis=1)
    assert np.abs(W_out_sh - edge.weights.detach().numpy()).max() < 1e-6
    for p in parts:
        assert np.abs(p["loss"] - obs.to_numpy("loss")).max() < 1e-6
        assert np.abs(p["out"] - obs.to_numpy("out")).max() < 1e-6
    assert np.abs(W_sh - _data()[0].numpy()).max() > 1e-7, "training did not change the weights"


def _inp_build(cls, where):
    """inp (3) --trained--> rnn (tanh, trained rows) --trained--> out (2)"""
    W, params, v0, x, W_in, W_out, u = _net_data()
    tgt = (torch.randn(T, 2, generator=torch.Generator().manual_seed(11)) * 0.1).numpy()
    net = cls(1e-2, **where)
    net.add_func_node("inp", 3, "identity")
    node = net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                               input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                               train_params=["weights"],
                               node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                          "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
    net.add_func_node("out", 2, "identity")
    e_in = net.add_edge("inp", "rnn", weights=W_in.numpy(), train="gd")
    e_out = net.add_edge("rnn", "out", weights=W_out.numpy(), train="gd")
    return net, node, e_in, e_out, u.numpy(), tgt


def _inp_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork
        net, node, e_in, e_out, u, tgt = _inp_build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()))
        net.fit_bptt(u, tgt, optimizer="sgd", lr=5e-2, update_steps=15, sampling_steps=1, verbose=False)
        np.savez(os.path.join(out_dir, f"inp_{rank}.npz"), W=node["weights"].detach().numpy(),
                 W_in=e_in.weights.detach().numpy(), W_out=e_out.weights.detach().numpy())
        try:  # a trained edge between replicated nodes upstream of the shard would need a gradient all-reduce
            net.add_func_node("pre", 3, "identity")
            net.add_edge("pre", "inp", weights=np.eye(3), train="gd")
            raise AssertionError("expected NotImplementedError")
        except NotImplementedError:
            pass
    finally:
        dist.destroy_process_group()


def test_sharded_fit_bptt_trains_the_input_edge(tmp_path, oracle_backend):
    """The trained input edge of a sharded population: each rank holds the rows of W_in of its neurons; their gradient
    is local (dL/dI_ext of the rank's neurons times the shared input)."""
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_inp_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    net, node, e_in, e_out, u, tgt = _inp_build(Network, dict(device="cpu", backend=oracle_backend))
    W_in0 = e_in.weights.detach().numpy().copy()
    net.fit_bptt(u, tgt, optimizer="sgd", lr=5e-2, update_steps=15, sampling_steps=1, verbose=False)
    parts = [np.load(tmp_path / f"inp_{r}.npz") for r in range(WORLD)]
    W_in_sh = np.concatenate([p["W_in"] for p in parts], axis=0)
    assert np.abs(W_in_sh - e_in.weights.detach().numpy()).max() < 1e-6
    assert np.abs(W_in_sh - W_in0).max() > 1e-6, "training did not change the input weights"
    assert np.abs(np.concatenate([p["W"] for p in parts], axis=0) - node["weights"].detach().numpy()).max() < 1e-6
    assert np.abs(np.concatenate([p["W_out"] for p in parts], axis=1) - e_out.weights.detach().numpy()).max() < 1e-6


QIF = "neuron_model_templates.spiking_neurons.qif.qif"


def _qif_build(cls, where):
    """QIF population (spikes within the window) with trained rows of W, eta, tau_s and a trained read-out."""
    W, params, v0, x = _data()
    g = torch.Generator().manual_seed(21)
    W_out = (torch.randn(2, N, generator=g) / N ** 0.5).numpy()
    tgt = (torch.randn(T, 2, generator=g) * 0.1).numpy()
    net = cls(1e-2, **where)
    node = net.add_diffeq_node("rnn", QIF, weights=(W.abs() * 2).numpy(), source_var="qif_op/s", target_var="qif_op/s_in",
                               input_var="qif_op/I_ext", output_var="qif_op/s", spike_var="qif_op/spike",
                               reset_var="qif_op/v", spike_threshold=5.0, spike_reset=-5.0, dtype=torch.float32,
                               train_params=["weights", "qif_op/eta", "qif_op/tau_s"],
                               node_vars={"all/qif_op/eta": (-5.0 + params["eta"]).numpy(), "all/qif_op/v": v0.numpy()})
    net.add_func_node("out", 2, "identity")
    edge = net.add_edge("rnn", "out", weights=W_out, train="gd")
    return net, node, edge, (x[:, :1] + 15.0).numpy().copy(), tgt


def _qif_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork
        net, node, edge, x, tgt = _qif_build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()))
        node.enable_spike_recording(4096)
        obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=15, sampling_steps=1, verbose=False)
        np.savez(os.path.join(out_dir, f"qif_bptt_{rank}.npz"), W=node["weights"].detach().numpy(),
                 eta=node["qif_op/eta"].detach().numpy(), tau_s=node["qif_op/tau_s"].detach().numpy(),
                 W_out=edge.weights.detach().numpy(), loss=obs.to_numpy("loss"), spikes=len(node.spikes()))
    finally:
        dist.destroy_process_group()


def test_sharded_spiking_fit_bptt_matches_single_process(tmp_path, oracle_backend):
    """Surrogate-gradient BPTT of a row-sharded QIF population on 2 gloo ranks == single-process fit_bptt."""
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_qif_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    net, node, edge, x, tgt = _qif_build(Network, dict(device="cpu", backend=oracle_backend))
    W0 = node["weights"].detach().numpy().copy()
    obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=15, sampling_steps=1, verbose=False)
    parts = [np.load(tmp_path / f"qif_bptt_{r}.npz") for r in range(WORLD)]
    assert sum(int(p["spikes"]) for p in parts) > 0, "the window must contain spikes"
    W_sh = np.concatenate([p["W"] for p in parts], axis=0)
    scale = np.abs(W_sh - W0).max()
    assert scale > 0, "training did not change the weights"
    assert np.abs(W_sh - node["weights"].detach().numpy()).max() < 1e-5 * max(1.0, scale / 1e-2)
    for key, name in (("eta", "qif_op/eta"), ("tau_s", "qif_op/tau_s")):
        got = np.concatenate([p[key] for p in parts])
        assert np.abs(got - node[name].detach().numpy()).max() < 2e-5, key
    W_out_sh = np.concatenate([p["W_out"] for p in parts], axis=1)
    assert np.abs(W_out_sh - edge.weights.detach().numpy()).max() < 1e-5
    for p in parts:
        assert np.abs(p["loss"] - obs.to_numpy("loss")).max() < 1e-5 * max(1.0, np.abs(obs.to_numpy("loss")).max())


SIGMOID = "neuron_model_templates.rate_neurons.leaky_integrator.sigmoid"


def _sig_build(cls, where):
    W, params, v0, x = _data()
    g = torch.Generator().manual_seed(31)
    act = dict(r_max=torch.rand(N, generator=g) + 0.5, s=torch.rand(N, generator=g) * 2 + 0.5, v0=torch.randn(N, generator=g) * 0.2)
    W_out = (torch.randn(2, N, generator=g) / N ** 0.5).numpy()
    tgt = (torch.randn(T, 2, generator=g) * 0.1).numpy()
    net = cls(1e-2, **where)
    nv = {f"all/li_op/{q}": params[q].numpy() for q in ("tau", "k", "eta")}
    nv.update({f"all/sigmoid_op/{q}": act[q].numpy() for q in act})
    nv["all/li_op/v"] = v0.numpy()
    node = net.add_diffeq_node("rnn", SIGMOID, weights=W.numpy(), source_var="sigmoid_op/r", target_var="li_op/r_in",
                               input_var="li_op/I_ext", output_var="sigmoid_op/r", dtype=torch.float32, node_vars=nv,
                               train_params=["weights", "sigmoid_op/r_max", "sigmoid_op/s", "sigmoid_op/v0"])
    net.add_func_node("out", 2, "identity")
    edge = net.add_edge("rnn", "out", weights=W_out, train="gd")
    return net, node, edge, x[:, :1].numpy().copy(), tgt


def _sig_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedNetwork
        net, node, edge, x, tgt = _sig_build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()))
        net.fit_bptt(x, tgt, optimizer="sgd", lr=5e-2, update_steps=15, sampling_steps=1, verbose=False)
        np.savez(os.path.join(out_dir, f"sig_{rank}.npz"), W=node["weights"].detach().numpy(),
                 **{q: node[f"sigmoid_op/{q}"].detach().numpy() for q in ("r_max", "s", "v0")})
    finally:
        dist.destroy_process_group()


def test_sharded_fit_bptt_trains_the_sigmoid_parameters(tmp_path, oracle_backend):
    from rectipy_amd import Network
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sig_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    net, node, edge, x, tgt = _sig_build(Network, dict(device="cpu", backend=oracle_backend))
    before = {q: node[f"sigmoid_op/{q}"].detach().numpy().copy() for q in ("r_max", "s", "v0")}
    net.fit_bptt(x, tgt, optimizer="sgd", lr=5e-2, update_steps=15, sampling_steps=1, verbose=False)
    parts = [np.load(tmp_path / f"sig_{r}.npz") for r in range(WORLD)]
    assert np.abs(np.concatenate([p["W"] for p in parts], axis=0) - node["weights"].detach().numpy()).max() < 1e-6
    for q in ("r_max", "s", "v0"):
        got = np.concatenate([p[q] for p in parts])
        assert np.abs(got - node[f"sigmoid_op/{q}"].detach().numpy()).max() < 2e-6, q
        assert np.abs(got - before[q]).max() > 1e-7, f"{q} was not trained"


def test_random_sharded_call_sequences_match_single_process():
    """scripts/fuzz_sharded.py: ShardedNetwork against the unsharded Network through random run / reset / state write /
    forward() / fit_bptt (sgd, adam) sequences on tanh and QIF populations, two gloo ranks (measured worst 9.5e-7 over 130
    sequences on 2 and 4 ranks)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_sharded
    assert fuzz_sharded.main(12, 1, 2) == 0
