"""Host-side cost of one Network.run call on the GPU box (N = 512, 20 steps): wall time per call and a cProfile of 300 calls."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import Network
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
net = Network(dt=1e-2, device="cuda:0")
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext", output_var="li_op/v",
                    weights=(torch.randn(N, N, device="cuda:0") / 200), source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
x = torch.zeros(3000, 1, device="cuda:0")
net.run(x[:5], sampling_steps=100, verbose=False, enable_grad=False)
torch.cuda.synchronize()
for steps in (1, 20):
    t0 = time.perf_counter()
    for i in range(300):
        net.run(x[5:5 + steps], sampling_steps=100, verbose=False, enable_grad=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"steps={steps}: host time per call {1e6 * (t1 - t0) / 300:.1f} us, incl. GPU drain {1e6 * (time.perf_counter() - t0) / 300:.1f} us")
pr = cProfile.Profile(); pr.enable()
for i in range(300):
    net.run(x[5:6], sampling_steps=100, verbose=False, enable_grad=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:4500])
