"""Child process of tests/test_rccl_inlib.py: a one-rank RCCL group on cuda:0.  Checks that the in-library exchange
(rp_forward_sharded / rp_backward_sharded, RCCL calls enqueued by the C loop) gives the same bits as the
unsharded library call and the same results as the torch.distributed per-step path.  Prints RCCL_INLIB_OK."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from rectipy_amd import Network, _abi
    from rectipy_amd.engine import Population
    from rectipy_amd.sharded import ShardedNetwork, ShardedPopulation

    N, T = 1024, 60
    g = torch.Generator().manual_seed(0)
    W = torch.randn(N, N, generator=g) * 1.5 / N ** 0.5
    params = dict(tau=torch.rand(N, generator=g) * 5 + 5, k=torch.rand(N, generator=g) + 0.5,
                  eta=torch.randn(N, generator=g) * 0.1)
    v0 = torch.randn(N, generator=g)
    x = (torch.randn(T, N, generator=g) * 0.5).to(dev)

    # ---- engine level: in-library all-gather == plain rp_forward, bit for bit -------------------------------
    sp = ShardedPopulation("li_tanh", W, params, 1e-2, dev, n=N, v0_local=v0)
    assert sp.comm is not None, "RCCL communicator was not picked up"
    full = Population("li_tanh", W, params, 1e-2, dev, v0=v0)
    traj_s, traj_f = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
    samp_s, samp_f = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
    r_full = torch.zeros(T, N, device=dev)
    n_s = sp.forward(T, x=x, traj=traj_s, out_var=_abi.RP_OUT_SOURCE, cutoff=3, sampling_steps=4, samples=samp_s,
                     save_src_full=r_full)
    n_f = full.forward(T, x=x, traj=traj_f, out_var=_abi.RP_OUT_SOURCE, cutoff=3, sampling_steps=4, samples=samp_f)
    torch.cuda.synchronize()
    assert n_s == n_f
    assert torch.equal(traj_s, traj_f) and torch.equal(samp_s, samp_f) and torch.equal(sp.pop.v, full.v)
    assert torch.equal(sp.pop.source, full.source)
    assert torch.equal(r_full, traj_f), "save_src_full must hold the source vector each step read"

    # ---- packed sparse shard + in-library exchange == dense unsharded ------------------------------------------
    Ns = 4096
    Ws = torch.randn(Ns, Ns, generator=g) * (1.5 / (0.1 * Ns) ** 0.5) * (torch.rand(Ns, Ns, generator=g) < 0.1)
    sps = ShardedPopulation("li_tanh", Ws, dict(tau=8.0, k=1.0, eta=0.0), 1e-2, dev, n=Ns)
    assert sps.pop.pack_weights().row_ids is None and sps.comm is not None
    dense = Population("li_tanh", Ws, dict(tau=8.0, k=1.0, eta=0.0), 1e-2, dev)
    xs = x[:, :1].contiguous()
    ta, tb = torch.zeros(T, Ns, device=dev), torch.zeros(T, Ns, device=dev)
    sps.forward(T, x=xs, traj=ta)
    dense.forward(T, x=xs, traj=tb)
    torch.cuda.synchronize()
    assert torch.equal(ta, tb) and torch.equal(sps.pop.v, dense.v), "packed sparse shard differs from the dense run"

    # ---- front end: ShardedNetwork.fit_bptt with the in-library exchange vs the torch.distributed path ------
    W_out = torch.randn(2, N, generator=g) / N ** 0.5
    u = torch.randn(40, 1, generator=g) * 0.5
    tgt = torch.randn(40, 2, generator=g) * 0.1

    def fit(cls, host_collectives):
        os.environ["RECTIPY_AMD_HOST_COLLECTIVES"] = "1" if host_collectives else "0"
        net = cls(1e-2, device="cuda:0")
        node = net.add_diffeq_node("rnn", TANH, weights=W.numpy(), source_var="tanh_op/r", target_var="li_op/r_in",
                            input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                            train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"],
                            node_vars={"all/li_op/tau": params["tau"].numpy(), "all/li_op/k": params["k"].numpy(),
                                       "all/li_op/eta": params["eta"].numpy(), "all/li_op/v": v0.numpy()})
        net.add_func_node("out", 2, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.numpy(), train="gd")
        if cls is ShardedNetwork:
            assert (net.comm is None) == host_collectives
        net.fit_bptt(u.numpy(), tgt.numpy(), optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=1, verbose=False)
        torch.cuda.synchronize()
        out = {k: node[f"li_op/{k}"].detach().cpu().numpy().copy() for k in ("tau", "k", "eta")}
        out["W"] = node["weights"].detach().cpu().numpy().copy()
        out["W_out"] = edge.weights.detach().cpu().numpy().copy()
        return out

    a = fit(ShardedNetwork, False)
    b = fit(ShardedNetwork, True)
    c = fit(Network, True)
    for key in a:
        assert np.array_equal(a[key], b[key]), f"in-library vs torch.distributed exchange differ in {key}"
        scale = np.abs(c[key]).max()
        assert np.abs(a[key] - c[key]).max() <= 2e-6 * scale + 1e-7, f"sharded vs unsharded fit differ in {key}"
    assert np.abs(a["W"] - W.numpy()).max() > 0, "training did not change the weights"

    # ---- spiking population: surrogate-gradient BPTT, in-library exchange vs torch.distributed vs unsharded ----
    QIF = "neuron_model_templates.spiking_neurons.qif.qif"
    Wq = (W.abs() * 2).numpy()
    uq = (u + 15.0).numpy()

    def fit_qif(cls, host_collectives):
        os.environ["RECTIPY_AMD_HOST_COLLECTIVES"] = "1" if host_collectives else "0"
        net = cls(1e-2, device="cuda:0")
        node = net.add_diffeq_node("rnn", QIF, weights=Wq, source_var="qif_op/s", target_var="qif_op/s_in",
                                   input_var="qif_op/I_ext", output_var="qif_op/s", spike_var="qif_op/spike",
                                   reset_var="qif_op/v", spike_threshold=5.0, spike_reset=-5.0, dtype=torch.float32,
                                   train_params=["weights", "qif_op/eta"],
                                   node_vars={"all/qif_op/eta": (-5.0 + params["eta"]).numpy(), "all/qif_op/v": v0.numpy()})
        net.add_func_node("out", 2, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.numpy(), train="gd")
        node.enable_spike_recording(1 << 16)
        net.fit_bptt(uq, tgt.numpy(), optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=1, verbose=False)
        torch.cuda.synchronize()
        assert len(node.spikes()) > 0, "the QIF window must contain spikes"
        return {"W": node["weights"].detach().cpu().numpy().copy(), "eta": node["qif_op/eta"].detach().cpu().numpy().copy(),
                "W_out": edge.weights.detach().cpu().numpy().copy()}

    a, b, c = fit_qif(ShardedNetwork, False), fit_qif(ShardedNetwork, True), fit_qif(Network, True)
    for key in a:
        assert np.array_equal(a[key], b[key]), f"QIF: in-library vs torch.distributed exchange differ in {key}"
        assert np.abs(a[key] - c[key]).max() <= 1e-5 * max(1.0, np.abs(c[key]).max()), f"QIF sharded vs unsharded: {key}"
    assert np.abs(a["W"] - Wq).max() > 0
    dist.destroy_process_group()
    print("RCCL_INLIB_OK", flush=True)


if __name__ == "__main__":
    main()
