"""Host time of ONE resident rp_forward call (100 steps, N = 512): the Python wrapper around it, the ctypes call itself, and the same
with a launch per step -- the host returns long before the GPU is done, so this is what a fit window pays on the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.engine import HipBackend, Population  # noqa: E402

dev = torch.device("cuda:0")
N, T, calls = 512, 100, 300
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) / N ** 0.5
x = torch.randn(T, N, device=dev, generator=gen) * 0.5
inner = {}
real = HipBackend.forward


def timed(self, *a, **kw):
    t0 = time.perf_counter()
    try:
        return real(self, *a, **kw)
    finally:
        inner["c"] = inner.get("c", 0.0) + time.perf_counter() - t0


HipBackend.forward = timed
for resident in ("1", "0"):
    os.environ["RECTIPY_AMD_RESIDENT"] = resident
    pop = Population("li_tanh", W, dict(tau=1.0, k=1.0, eta=0.0), 1e-2, dev)
    sv, ss = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
    for _ in range(20):
        pop.forward(T, x=x, save_v=sv, save_src=ss)
    torch.cuda.synchronize()
    inner.clear()
    t0 = time.perf_counter()
    for _ in range(calls):
        pop.forward(T, x=x, save_v=sv, save_src=ss)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"resident={resident}: per call host {host / calls * 1e6:.1f} us (ctypes call {inner['c'] / calls * 1e6:.1f} us), wall {wall / calls * 1e6:.1f} us")
