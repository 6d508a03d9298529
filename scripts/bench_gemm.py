"""dW GEMM (rp_outer_accum) timing at the BPTT window shape: N = 32768, T = 1000 (and a bit-exactness spot check
against the oracle order at a size where the XCD-aware tile order is active)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd.engine import HipBackend  # noqa: E402

dev = torch.device("cuda:0")
be = HipBackend()
for N, T in ((8192, 1000), (16384, 1000), (32768, 1000)):
    A = torch.randn(T, N, device=dev) * 0.01
    B = torch.tanh(torch.randn(T, N, device=dev))
    Cm = torch.zeros(N, N, device=dev)
    be.outer_accum(N, T, 1e-2, A.data_ptr(), N, B.data_ptr(), N, Cm.data_ptr(), N, dev)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        be.outer_accum(N, T, 1e-2, A.data_ptr(), N, B.data_ptr(), N, Cm.data_ptr(), N, dev)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    # spot check of 64 entries against a float64 product (4 accumulations so far)
    idx = torch.randint(0, N, (64, 2), generator=torch.Generator().manual_seed(0))
    ref = torch.stack([(A[:, i].double() * B[:, j].double()).sum() * 1e-2 * 4 for i, j in idx.tolist()])
    got = torch.stack([Cm[i, j].double() for i, j in idx.tolist()])
    print(json.dumps(dict(N=N, T=T, ms=round(best, 3), TFLOPs=round(2.0 * N * N * T / best / 1e9, 1),
                          max_rel_err=float(((got - ref).abs() / ref.abs().clamp_min(1e-9)).max()))), flush=True)
    del A, B, Cm
