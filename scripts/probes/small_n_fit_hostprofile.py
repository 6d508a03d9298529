import contextlib, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rectipy_amd import Network
N, T = 512, 10000
dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
x = torch.randn(T, N, device=dev, generator=gen) * 0.5
target = torch.randn(T, 3, device=dev, generator=gen) * 0.1
with contextlib.redirect_stdout(io.StringIO()):
    fit = Network(dt=1e-2, device=str(dev))
    fit.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W * 0.5, source_var="tanh_op/r",
                        target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
    fit.add_func_node("out", 3, "identity")
    fit.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="gd")
    fit.fit_bptt(x[:200], target[:200], optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print("per window: host %.0f us, wall %.0f us" % (t_host / 100 * 1e6, (time.perf_counter() - t0) / 100 * 1e6))
pr = cProfile.Profile(); pr.enable()
fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38); print(s.getvalue()[:7000])
