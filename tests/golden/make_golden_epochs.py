"""G30: Network.fit_bptt with LIST inputs (one optimizer step per epoch on the sampled outputs of `run`, state restored
after every epoch; reference rectipy/network.py:983-1014), SGD with momentum passed through `optimizer_kwargs`,
sampling_steps = 2, three epochs, recurrent weights + per-neuron parameters + gd read-out trained.
Run in the build container only:  python tests/golden/make_golden_epochs.py"""
import numpy as np
import torch

from make_golden import import_reference, np_, rate_node, rng_li, save


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    n, kout, T, dt, s, lr = 32, 2, 60, 1e-2, 2, 1e-2
    P = rng_li(n, seed=301)
    r = np.random.default_rng(302)
    W_out = r.standard_normal((kout, n)) / np.sqrt(n)
    xs = [r.standard_normal((T, n)) * 0.5 for _ in range(3)]
    targets = [r.standard_normal((T // s, kout)) * 0.1 for _ in range(3)]
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        net = Network(dt, dtype=dtype)
        node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype,
                         train_params=["weights", "li_op/tau", "li_op/eta"])
        net.add_node("rnn", node, node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        obs = net.fit_bptt([torch.tensor(x, dtype=dtype) for x in xs], targets, optimizer="sgd", lr=lr,
                           optimizer_kwargs={"momentum": 0.9}, sampling_steps=s, verbose=False)
        out["epoch_loss_" + tag] = np.asarray(obs["epoch_loss"])
        out["W_after_" + tag] = np_(node["weights"])
        out["tau_after_" + tag] = np_(node["li_op/tau"])
        out["eta_after_" + tag] = np_(node["li_op/eta"])
        out["W_out_after_" + tag] = np_(edge.weights)
        out["state_" + tag] = np_(net.state["rnn"])
        out["last_out_" + tag] = obs.to_numpy("out")
        print(tag, out["epoch_loss_" + tag])
    save("g30_bptt_epochs", dt=dt, lr=lr, sampling_steps=s, xs=np.stack(xs), targets=np.stack(targets), W_out=W_out, **P, **out)


if __name__ == "__main__":
    main()
