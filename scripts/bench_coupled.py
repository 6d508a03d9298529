"""Time the coupled-population step (rp_forward_coupled: two tanh populations of N/2 neurons, a -> b and feedback b -> a,
dense N/2 x N/2 edges) per walk of the edge kernel.  Measurement build: RECTIPY_AMD_MEASUREMENT=1 python scripts/bench_coupled.py"""
import json
import os
import sys

os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rectipy_amd import _abi, _lib  # noqa: E402
from rectipy_amd.engine import Population, forward_coupled  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = {}
    for N, T in ((1024, 4000), (2048, 4000), (4096, 2000), (8192, 1000)):
        h = N // 2
        gen = torch.Generator(device=dev).manual_seed(N)
        mk = lambda g: torch.randn(h, h, device=dev, generator=gen) * g / h ** 0.5
        pops = [Population("li_tanh", mk(1.0), dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev) for _ in range(2)]
        E_ab, E_ba = mk(0.5), mk(0.5)
        x = torch.randn(T, h, device=dev, generator=gen) * 0.1
        hist = [torch.zeros(h, device=dev) for _ in range(3)]

        def run(steps):
            forward_coupled([dict(pop=pops[0], run=dict(steps=steps, x=x[:steps], hist=hist), edges=[(E_ba, 1, _abi.RP_EDGE_LIVE)]),
                             dict(pop=pops[1], run=dict(steps=steps), edges=[(E_ab, 0, _abi.RP_EDGE_OUT)])])

        res = {}
        cfgs = [(0, 0, 0), (2, 1, 683), (2, 1, 512), (2, 1, 1024), (2, 1, 0), (4, 2, 0), (4, 2, 2048), (4, 2, 512), (2, 2, 0), (1, 4, 0),
                (8, 1, 0), (8, 1, 512), (2, 2, 683), (1, 8, 0)]
        for R, U, blocks in cfgs:
            lib.rp_set_tuning(R, U, 1, blocks)
            run(max(2, T // 10))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(T)
            e1.record()
            torch.cuda.synchronize()
            res[f"R{R}_U{U}_wg{blocks}"] = round(e0.elapsed_time(e1) / T * 1e3, 2)
        lib.rp_set_tuning(0, 0, 1, 0)
        out[f"N{N}"] = res
        print(N, res, flush=True)
        del pops, E_ab, E_ba
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
