"""ms per step of the matrix-core batched forms at N (default 32768): the 16x16x4 LDS-DMA form over (tile columns, ring
stages) x batch, against the r02 32x32x2 form and the vector-ALU form (GPU box).  Interleaved rounds in one process."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ROUNDS = 3
lib = _lib.load()
W = bench.gen_rows(N, 0, N, dev)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cases = []
for B in (8, 16, 32):
    cases.append(("mfma32", B, 0, 0, 0))
    for tc, st, wv in ((64, 2, 4), (64, 3, 4), (128, 2, 4)) + (((64, 2, 8), (64, 3, 8), (128, 2, 8)) if B > 16 else ()):
        cases.append(("mfma16", B, tc, st, wv))
if os.environ.get("SIZES"):  # the candidates for the dispatch rule, at another N
    cases = [("valu", 8, 0, 0, 0), ("mfma16", 8, 64, 2, 4), ("mfma32", 16, 0, 0, 0), ("mfma16", 16, 64, 2, 4), ("mfma16", 16, 64, 3, 4),
             ("mfma32", 32, 0, 0, 0), ("mfma16", 32, 64, 3, 4), ("mfma16", 32, 64, 3, 8), ("mfma16", 32, 64, 3, 16)]
elif os.environ.get("SPLIT"):  # the role-split kernel (waves = 16), whole and with halves switched off (116 / 216)
    cases = [("mfma16", 32, 64, 3, 8), ("mfma16", 32, 64, 3, 16), ("mfma16", 32, 64, 4, 16), ("mfma16", 32, 64, 2, 16),
             ("mfma16", 32, 64, 3, 116), ("mfma16", 32, 64, 3, 216), ("mfma16", 32, 64, 4, 116), ("mfma16", 17, 64, 3, 16),
             ("mfma16", 32, 64, 3, 1016), ("mfma16", 32, 64, 3, 2016), ("mfma16", 32, 64, 3, 2116)]
elif os.environ.get("PROBE_HALVES"):  # waves + 10 / 20: the kernel without its MFMAs / without its steady-state DMA
    cases = [("mfma16", 32, 64, 3, 4), ("mfma16", 32, 64, 3, 14), ("mfma16", 32, 64, 3, 24),
             ("mfma16", 32, 64, 3, 8), ("mfma16", 32, 64, 3, 18), ("mfma16", 32, 64, 3, 28), ("mfma16", 32, 64, 3, 34), ("mfma16", 32, 64, 3, 38)]
state = {}
for form, B, tc, st, wv in cases:
    bp = BatchedPopulation(pop, B, form=form)
    bp.v.normal_(generator=torch.Generator(device=dev).manual_seed(B))
    bp.refresh()
    state[(form, B, tc, st, wv)] = (bp, torch.randn(T + 10, B, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))
times = {k: [] for k in state}
for r in range(ROUNDS):
    for key, (bp, x) in state.items():
        lib.rp_set_batched_mfma_tuning(key[2], key[3], key[4])
        try:
            bp.forward(10, x=x[:10]); torch.cuda.synchronize()
            e0.record(); bp.forward(T, x=x[10:]); e1.record(); torch.cuda.synchronize()
            times[key].append(e0.elapsed_time(e1) / T)
        except Exception as err:  # noqa: BLE001 (an instance that does not fit the LDS)
            times[key].append(float("nan"))
            print(json.dumps({"case": key, "error": str(err)[:200]}), flush=True)
lib.rp_set_batched_mfma_tuning(0, 0, 0)
for key, ts in times.items():
    ms = sorted(ts)[len(ts) // 2]
    form, B, tc, st, wv = key
    print(json.dumps({"form": form, "B": B, "tile_cols": tc, "stages": st, "waves": wv, "ms_per_step_median": round(ms, 4), "ms_min": round(min(ts), 4),
                      "W_stream_TBps": round(4.0 * N * N / (ms * 1e-3) / 1e12, 2),
                      "fma_TFLOPs": round(2.0 * N * N * B / (ms * 1e-3) / 1e12, 1),
                      "finite": bool(torch.isfinite(state[key][0].v).all().item())}), flush=True)
