"""Row-sharded integration through the front end, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_run.py

Every rank builds the same graph; ShardedNetwork keeps this rank's rows of the recurrent matrix and exchanges the
source vector with one RCCL all-gather per Euler step (BASELINE.json configs[4] shape when N = 262144 on 8 GPUs)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gen_rows  # noqa: E402
from rectipy_amd.sharded import ShardedNetwork, one_rank_env, shard_bounds  # noqa: E402

N = int(os.environ.get("N", 32768))
T = int(os.environ.get("T", 1000))
local_rank = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
if "RANK" not in os.environ:  # started with plain `python`: one rank on this GPU
    one_rank_env()  # fills in only what is missing; free port
dist.init_process_group("nccl", device_id=dev)
row0, rows = shard_bounds(N, dist.get_world_size(), dist.get_rank())

net = ShardedNetwork(dt=1e-2, device=str(dev))
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                    output_var="tanh_op/r", weights=gen_rows(N, row0, rows, dev),   # this rank's row block, made on device
                    source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
net.add_func_node("out", 3, "identity")
net.add_edge("rnn", "out", weights=np.random.default_rng(0).standard_normal((3, N)) / np.sqrt(N))
inp = torch.sin(torch.arange(T, device=dev) * 0.0126).reshape(-1, 1)
obs = net.run(inp, sampling_steps=100, verbose=False, enable_grad=False)
if dist.get_rank() == 0:
    print("read-out samples:", obs.to_numpy("out").shape, "finite:", bool(np.isfinite(obs.to_numpy("out")).all()))
dist.destroy_process_group()
