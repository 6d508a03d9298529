"""Role-split batched step, B = 32, N = 32768: cache policy of the two LDS-DMA streams (measurement build, PROBE 30..37)."""
import os
import sys

os.environ["RECTIPY_AMD_MEASUREMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
N, B, T = 32768, 32, 100
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
bp = BatchedPopulation(pop, B, form="mfma16")
bp.v.normal_(generator=gen)
bp.refresh()
x = torch.randn(T + 10, B, 1, device=dev, generator=gen)
names = {16: "default (W nt, source plain)", 2016: "start 0", 2116: "one start per XCD", 3016: "W nt+sc1", 3116: "W sc0+nt+sc1", 3216: "W sc0+sc1",
         3316: "W plain", 3416: "W sc1", 3516: "source sc0", 3616: "source sc1", 3716: "W sc0+nt", 116: "no MFMAs", 216: "no DMA"}
for rep in range(2):
    for waves, label in names.items():
        lib.rp_set_batched_mfma_tuning(64, 3, waves)
        bp.forward(10, x=x[:10])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bp.forward(T, x=x[10:])
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:32s} {e0.elapsed_time(e1) / T:.4f} ms", flush=True)
        bp.v.normal_(generator=gen)
        bp.refresh()
lib.rp_set_batched_mfma_tuning(0, 0, 0)
