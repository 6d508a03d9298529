"""The adjoint's column-strip kernel (wt_partial_kernel) on one n x n matrix: us per launch for the built-in walk and for the
measurement build's (rows in flight, strip width, grid cap) settings.  Shapes: 16384^2 (a member of the coupled fit leg) and
32768^2 (the single population)."""
import json
import os
import sys

os.environ["RECTIPY_AMD_MEASUREMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for n in (16384, 32768):
    pop = Population("li_tanh", bench.gen_rows(n, 0, n, dev), dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    lib = pop.backend.lib
    T = 4
    f32 = dict(dtype=torch.float32, device=dev)
    tv, tr, u = torch.zeros(T, n, **f32), torch.zeros(T, n, **f32), torch.randn(T, n, **f32)
    lam = torch.zeros(n, **f32)
    net, b, keep = pop._backward_args(traj_v=tv, traj_r=tr, grad_out=None, grad_var=_abi.RP_OUT_STATE0, lam=lam, u_traj=u,
                                      _args_only="stepwise")
    z = torch.empty(n, **f32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(reps=60):
        for _ in range(5):
            pop.backend.bwd_partial(net, b, 1, z.data_ptr(), dev)
        e0.record()
        for _ in range(reps):
            pop.backend.bwd_partial(net, b, 1, z.data_ptr(), dev)
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps * 1e3, 1)   # (includes the 4.7 us chunk-sum launch)

    res = {"built_in": timed()}
    for U, SW in ((2, 1), (1, 1), (3, 1), (4, 1), (8, 1), (2, 2), (4, 2), (1, 2)):
        for cap in (0, 512, 640, 768, 1024, 1536, 2048):
            lib.rp_set_bwd_tuning(U, SW)
            lib.rp_set_bwd_grid(cap)
            res[f"U{U}_SW{SW}_cap{cap}"] = timed(30)
    lib.rp_set_bwd_tuning(0, 0)
    lib.rp_set_bwd_grid(0)
    out[f"n{n}"] = dict(sorted(res.items(), key=lambda kv: kv[1])[:12], built_in=res["built_in"])
    print(out, file=sys.stderr, flush=True)
    del pop
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
