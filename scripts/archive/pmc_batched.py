"""Workload for the PMC passes over the batched matrix-core step (scripts/archive/profile_batched.sh): a few dozen steps of each
form / configuration at N = 32768, so that every template instance appears under its own kernel name in the trace."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402

dev = torch.device("cuda:0")
N, T = 32768, 40
lib = _lib.load()
W = bench.gen_rows(N, 0, N, dev)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
cases = [("mfma32", 32, 0, 0, 0), ("mfma16", 8, 64, 3, 4), ("mfma16", 16, 64, 3, 4), ("mfma16", 32, 64, 3, 4),
         ("mfma16", 32, 64, 3, 16)]
cases += [tuple([c[0], int(c[1]), int(c[2]), int(c[3]), int(c[4])]) for c in (a.split(",") for a in sys.argv[1:])]
for form, B, tc, st, wv in cases:
    lib.rp_set_batched_mfma_tuning(tc, st, wv)
    bp = BatchedPopulation(pop, B, form=form)
    bp.v.normal_(generator=torch.Generator(device=dev).manual_seed(B))
    bp.refresh()
    x = torch.randn(T, B, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    bp.forward(T, x=x)
    torch.cuda.synchronize()
lib.rp_set_batched_mfma_tuning(0, 0, 0)
