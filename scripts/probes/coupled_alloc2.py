"""Repeat: four separate 1 GiB allocations vs one arena for the matrices of the coupled step, interleaved trials in one
process (allocator emptied in between)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population, forward_coupled  # noqa: E402

dev = torch.device("cuda:0")
T, N = 100, 32768
h = N // 2
gen = torch.Generator(device=dev).manual_seed(1)
xh = torch.randn(T, h, device=dev, generator=gen) * 0.1


def timed(mats, label):
    pops = [Population("li_tanh", mats[q], dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev) for q in range(2)]
    hist = [torch.zeros(h, device=dev) for _ in range(3)]

    def run(steps):
        forward_coupled([dict(pop=pops[0], run=dict(steps=steps, x=xh[:steps], hist=hist), edges=[(mats[3], 1, _abi.RP_EDGE_LIVE)]),
                         dict(pop=pops[1], run=dict(steps=steps), edges=[(mats[2], 0, _abi.RP_EDGE_OUT)])])
    run(10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(T)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:28s} {e0.elapsed_time(e1) / T * 1e3:7.1f} us", [hex(m.data_ptr())[-9:] for m in mats], flush=True)


for trial in range(4):
    mats = [torch.empty(h, h, device=dev).normal_(generator=gen).mul_(1 / h ** 0.5) for _ in range(4)]
    timed(mats, f"separate #{trial}")
    # the same matrices, members swapped: launch 1 streams (m1, m2), launch 2 (m0, m3)
    timed([mats[1], mats[0], mats[3], mats[2]], f"separate #{trial} swapped")
    del mats
    torch.cuda.empty_cache()
    arena = torch.empty(4 * h * h + trial * 524288 * 3, device=dev).normal_(generator=gen).mul_(1 / h ** 0.5)
    mats = [arena[q * h * h:(q + 1) * h * h].view(h, h) for q in range(4)]
    timed(mats, f"arena #{trial}")
    del mats, arena
    torch.cuda.empty_cache()
    # separate allocations of 1 GiB + 2 MiB * k each (other offsets between them)
    mats = [torch.empty(h * h + 524288 * (q + 1 + trial), device=dev)[:h * h].view(h, h).normal_(generator=gen).mul_(1 / h ** 0.5) for q in range(4)]
    timed(mats, f"separate padded #{trial}")
    del mats
    torch.cuda.empty_cache()
