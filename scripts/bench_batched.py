"""Trajectory-steps/s of the batched step (B trajectories share one pass over W) at N = 32768 (GPU box)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 300
W = bench.gen_rows(N, 0, N, dev)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
x1 = torch.zeros(T, 1, device=dev)
pop.forward(20, x=x1[:20]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); pop.forward(T, x=x1); e1.record(); torch.cuda.synchronize()
ms1 = e0.elapsed_time(e1) / T
print(json.dumps({"B": 1, "ms_per_step": round(ms1, 4), "trajectory_steps_per_s": round(1e3 / ms1, 1)}), flush=True)
from rectipy_amd import _lib  # noqa: E402
lib = _lib.load()
for B, R, U in ((2, 0, 0), (4, 0, 0), (4, 8, 1), (4, 8, 2), (4, 4, 1), (8, 0, 0), (8, 2, 2), (8, 2, 4), (8, 4, 2), (16, 0, 0), (16, 2, 1),
                (16, 1, 4), (16, 1, 2)):
    lib.rp_set_batched_tuning(R, U)
    bp = BatchedPopulation(pop, B)
    bp.v.normal_()
    bp.refresh()
    x = torch.randn(T, B, 1, device=dev)
    bp.forward(20, x=x[:20]); torch.cuda.synchronize()
    e0.record(); bp.forward(T, x=x); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / T
    print(json.dumps({"B": B, "R": R, "U": U, "ms_per_step": round(ms, 4), "trajectory_steps_per_s": round(B * 1e3 / ms, 1),
                      "speedup_vs_B_unbatched_runs": round(B * ms1 / ms, 2),
                      "W_stream_TBps": round(4.0 * N * N / (ms * 1e-3) / 1e12, 2),
                      "fma_TFLOPs": round(2.0 * N * N * B / (ms * 1e-3) / 1e12, 1),
                      "finite": bool(torch.isfinite(bp.v).all().item())}), flush=True)

for B, U in ((32, 0), (16, 0)):
    lib.rp_set_batched_tuning(0, U)
    bp = BatchedPopulation(pop, B, mfma=True)
    bp.v.normal_()
    bp.refresh()
    x = torch.randn(T, B, 1, device=dev)
    bp.forward(20, x=x[:20]); torch.cuda.synchronize()
    e0.record(); bp.forward(T, x=x); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / T
    print(json.dumps({"form": "mfma", "B": B, "ms_per_step": round(ms, 4), "trajectory_steps_per_s": round(B * 1e3 / ms, 1),
                      "speedup_vs_B_unbatched_runs": round(B * ms1 / ms, 2),
                      "W_stream_TBps": round(4.0 * N * N / (ms * 1e-3) / 1e12, 2),
                      "mfma_TFLOPs_32_slots": round(2.0 * N * N * 32 / (ms * 1e-3) / 1e12, 1),
                      "finite": bool(torch.isfinite(bp.v).all().item())}), flush=True)
