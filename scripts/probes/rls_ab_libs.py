import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from rectipy_amd.engine import HipBackend
from rectipy_amd.rls import RLS
dev = torch.device("cuda:0")
n_in, k, T = 32768, 3, 64
gen = torch.Generator(device=dev).manual_seed(5)
X = torch.randn(T, n_in, device=dev, generator=gen) * 0.1
Y = torch.randn(T, k, device=dev, generator=gen)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for name, be in (("product", None), ("measure", HipBackend(measurement=True)), ("product2", None)):
    edge = (RLS(n_in, k, beta=0.99, alpha=1.0, backend=be) if be is not None else RLS(n_in, k, beta=0.99, alpha=1.0)).to(dev)
    edge.symmetric, edge.block = True, 8
    for rep in range(3):
        edge.sweep(X[:16], Y[:16], 1)
        torch.cuda.synchronize()
        e0.record()
        edge.sweep(X, Y, 1)
        e1.record()
        torch.cuda.synchronize()
        out[f"{name}_{rep}"] = round(e0.elapsed_time(e1) / T, 4)
    del edge
    torch.cuda.empty_cache()
print(json.dumps(out))
