"""Where do the ~110 ms of a process's first Network.run go? (GPU box)"""
import os, sys, time, json
t00 = time.perf_counter()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
marks = [("import torch + numpy", time.perf_counter() - t00)]
def mark(name, t0):
    torch.cuda.synchronize() if torch.cuda.is_initialized() else None
    marks.append((name, time.perf_counter() - t0))
t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda:0"); mark("torch cuda init + first tensor", t0)
t0 = time.perf_counter(); import rectipy_amd; from rectipy_amd import _lib, Network; mark("import rectipy_amd", t0)
t0 = time.perf_counter(); lib = _lib.load(); mark("dlopen library + ABI checks", t0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = (np.random.default_rng(0).standard_normal((N, N)) / np.sqrt(N)).astype(np.float32)
t0 = time.perf_counter()
net = Network(1e-2, device="cuda:0")
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", weights=W, source_var="tanh_op/r",
                    target_var="li_op/r_in", input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32)
mark("Network + add_diffeq_node (W upload)", t0)
x = np.zeros((100, N), dtype=np.float32)
for i in range(3):
    t0 = time.perf_counter(); net.run(x, sampling_steps=10, verbose=False, enable_grad=False); mark(f"Network.run #{i + 1} (100 steps)", t0)
pop = net.get_node("rnn")._pop
t0 = time.perf_counter(); pop.forward(5); mark("pop.forward(5) warm", t0)
print(json.dumps({k: round(v * 1e3, 2) for k, v in marks}, indent=1))
