"""Window losses of bench.py's facade fit_bptt leg for a few SGD learning rates (GPU box)."""
import contextlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rectipy_amd import Network  # noqa: E402
dev = torch.device("cuda:0")
N, T, win, dt = 32768, 4000, 1000, 1e-2
W0 = bench.gen_rows(N, 0, N, dev)
tt = torch.arange(T, device=dev, dtype=torch.float32) * dt
x = torch.sin(2 * torch.pi * 0.2 * tt).reshape(-1, 1).contiguous()
for lr in (1e-5, 1e-6, 1e-7):
    gen = torch.Generator(device=dev).manual_seed(3)
    net = Network(dt=dt, device=str(dev))
    net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                        output_var="li_op/v", weights=W0.clone(), source_var="tanh_op/r", target_var="li_op/r_in",
                        train_params=["weights"], dtype=torch.float32)
    net.add_func_node("out", 3, "identity")
    net.add_edge("rnn", "out", weights=torch.randn(3, N, generator=gen, device=dev) / N ** 0.5, train="gd")
    tgt = torch.randn(T, 3, generator=gen, device=dev) * 0.1
    with contextlib.redirect_stdout(sys.stderr):
        obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=lr, update_steps=win, sampling_steps=1, verbose=False)
    losses = obs["loss"]
    print(json.dumps({"lr": lr, "window_losses": [round(float(losses[b - 1]), 4) for b in range(win, T + 1, win)]}), flush=True)
    del net
    torch.cuda.empty_cache()
