"""Does the coupled step (two 16384-neuron populations, four 1 GiB matrices) depend on WHERE the matrices lie relative to
each other?  One arena, the four matrices carved out of it `pad` bytes apart; single-population N = 32768 step on the same
box as the yardstick.  Events around 200-step library calls, three repetitions."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population, forward_coupled  # noqa: E402

dev = torch.device("cuda:0")
T = 200


def timed(fn):
    fn(20)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(T)
        e1.record()
        torch.cuda.synchronize()
        best.append(round(e0.elapsed_time(e1) / T * 1e3, 1))
    return best


out = {}
N = 32768
gen = torch.Generator(device=dev).manual_seed(1)
W = torch.randn(N, N, device=dev, generator=gen) / N ** 0.5
pop = Population("li_tanh", W, dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev)
x = torch.randn(T, N, device=dev, generator=gen) * 0.1
out["single_population_N32768"] = timed(lambda s: pop.forward(s, x=x[:s]))
del pop, W, x
torch.cuda.empty_cache()

h = N // 2
mat = h * h
for pad in (0, 1024, 17 * 1024 + 256, 256 * 1024 + 1024, 4 * 1024 * 1024 + 17 * 1024):  # floats
    arena = torch.empty(4 * mat + 4 * pad + 1024, device=dev)
    views = []
    for q in range(4):
        off = q * (mat + pad)
        v = arena[off:off + mat].view(h, h)
        v.normal_(generator=gen)
        v.mul_((1.0 if q < 2 else 0.5) / h ** 0.5)
        views.append(v)
    pops = [Population("li_tanh", views[q], dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev) for q in range(2)]
    for q in range(2):
        assert pops[q].W.data_ptr() == views[q].data_ptr()
    xh = torch.randn(T, h, device=dev, generator=gen) * 0.1
    hist = [torch.zeros(h, device=dev) for _ in range(3)]

    def run(steps):
        forward_coupled([dict(pop=pops[0], run=dict(steps=steps, x=xh[:steps], hist=hist), edges=[(views[3], 1, _abi.RP_EDGE_LIVE)]),
                         dict(pop=pops[1], run=dict(steps=steps), edges=[(views[2], 0, _abi.RP_EDGE_OUT)])])
    out[f"coupled_pad_{pad * 4}B"] = timed(run)
    print(pad, out[f"coupled_pad_{pad * 4}B"], [hex(v.data_ptr()) for v in views], flush=True)
    del pops, views, arena
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
