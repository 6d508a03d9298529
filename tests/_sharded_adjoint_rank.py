"""One rank of tests/test_sharded_adjoint.py::test_direct_exchange_adjoint_on_one_gpu_matches_the_mirror: RANK / WORLD_SIZE
ranks share cuda:0, coordinate through a gloo group and run forward window + adjoint sweep of their row shard with
``exchange="direct"`` (twice).  Writes the rank's adjoint state and gradients to OUT_DIR."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = os.environ["OUT_DIR"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _sharded_adjoint_common import backward_kwargs, window_data
    from rectipy_amd import _abi
    from rectipy_amd.sharded import ShardedPopulation
    from test_sharded_adjoint import CASES
    for model, N, T in CASES:
        W, params, v0, x, g_out, kw = window_data(N, T, model)
        rows = N // world
        sl = slice(rank * rows, (rank + 1) * rows)
        if model == "qif":
            kw["s0_local"] = torch.zeros(rows)
        for rep in (0, 1):
            sp = ShardedPopulation(model, W[sl], {k: v[sl] for k, v in params.items()}, 1e-2, dev, n=N, v0_local=v0[sl],
                                   exchange="direct", **kw)
            assert sp.direct is not None and sp.comm.mode == _abi.RP_COMM_DIRECT and sp.comm.inbox
            saved = dict(v=torch.zeros(T, rows, device=dev), r=torch.zeros(T, rows, device=dev), acc=torch.zeros(T, rows, device=dev))
            xs = x[:, sl].contiguous().to(dev)
            if rep == 1:  # an odd number of exchanges first: the sweep then starts on the other inbox parity
                sp.forward(3)
                sp.pop.v.copy_(v0[sl].to(dev))
                if model == "qif":
                    sp.pop.set_source_local(torch.zeros(rows, device=dev))
                else:
                    sp.pop.refresh()
                torch.cuda.synchronize()
                dist.barrier()
                sp._gather_source()
                torch.cuda.synchronize()
                dist.barrier()
            sp.forward(T, x=xs, save_v=saved["v"], save_src=saved["r"], save_acc=saved["acc"])
            kw_b, res = backward_kwargs(model, sp.pop, T, dev, saved, g_out[:, sl].contiguous().to(dev), xs)
            args = dict(kw_b)
            keep = sp.backward(args.pop("traj_v"), args.pop("traj_r"), args.pop("grad_out"), args.pop("grad_var"), args.pop("lam"),
                               args.pop("u_traj"), **args)
            torch.cuda.synchronize()
            del keep
            np.savez(os.path.join(out_dir, f"{model}_{rank}_{rep}.npz"), **{k: v.cpu().numpy() for k, v in res.items()})
            dist.barrier()
            sp.direct.close()
            dist.barrier()
    dist.destroy_process_group()
    print(f"SHARDED_ADJOINT_RANK_{rank}_OK", flush=True)


if __name__ == "__main__":
    main()
