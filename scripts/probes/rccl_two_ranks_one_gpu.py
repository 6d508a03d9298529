"""Can two RCCL ranks share the one GPU of a gpurun box?  (Would let the >1-rank in-library exchange be rehearsed here.)
Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 <this>"""
import os
import sys

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev)
    t = torch.full((4,), float(dist.get_rank() + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {dist.get_rank()}: all_reduce on a shared GPU gave {t.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as err:  # noqa: BLE001
    print(f"rank {os.environ.get('RANK')}: RCCL refused two ranks on one device: {type(err).__name__}: {str(err)[:300]}", flush=True)
    sys.exit(3)
