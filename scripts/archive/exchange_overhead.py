"""Per-step cost of the source-vector exchange on the GPU box (one RCCL rank): in-library RCCL calls on the kernel
stream (rp_forward_sharded) vs torch.distributed.all_gather_into_tensor from a Python loop, vs no exchange at all.
With one rank the collective moves no data, so the differences are pure launch / hand-over overhead - the part of the
multi-GPU step that does not shrink with the shard."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

from rectipy_amd.sharded import ShardedPopulation  # noqa: E402

for N, T in ((2048, 4000), (4096, 4000), (8192, 3000), (32768, 600)):
    W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
    sp = ShardedPopulation("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=N)
    comm = sp.comm
    assert comm is not None
    x = torch.sin(torch.arange(T, device=dev) * 0.01).reshape(-1, 1).contiguous()
    res = {"N": N}
    for label in ("no_exchange", "in_library", "torch_distributed"):
        def run():
            if label == "no_exchange":
                sp.pop.forward(T, x=x)
            else:
                sp.comm = comm if label == "in_library" else None
                sp.forward(T, x=x)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        res[label + "_us_per_step"] = round((time.perf_counter() - t0) / T * 1e6, 2)
    print(json.dumps(res), flush=True)
    del sp, W
dist.destroy_process_group()
