"""Row-sharded BPTT through the front end, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_fit.py

Each rank trains its rows of the recurrent matrix and its columns of the read-out; per step the forward pass
all-gathers the source vector and the adjoint reduce-scatters W_p^T u_p (RCCL over xGMI)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gen_rows  # noqa: E402
from rectipy_amd.sharded import ShardedNetwork, one_rank_env, shard_bounds  # noqa: E402

N = int(os.environ.get("N", 8192))
T = int(os.environ.get("T", 400))
local_rank = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
if "RANK" not in os.environ:  # started with plain `python`: one rank on this GPU
    one_rank_env()  # fills in only what is missing; free port
dist.init_process_group("nccl", device_id=dev)
row0, rows = shard_bounds(N, dist.get_world_size(), dist.get_rank())

net = ShardedNetwork(dt=1e-2, device=str(dev))
node = net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                           output_var="li_op/v", weights=gen_rows(N, row0, rows, dev, g=0.9), source_var="tanh_op/r",
                           target_var="li_op/r_in", train_params=["weights"], dtype=torch.float32)
net.add_func_node("out", 3, "identity")
rng = np.random.default_rng(0)
net.add_edge("rnn", "out", weights=rng.standard_normal((3, N)) / np.sqrt(N), train="gd")
inp = torch.sin(torch.arange(T, device=dev) * 0.0126).reshape(-1, 1)
tgt = torch.tensor(rng.standard_normal((T, 3)) * 0.1, dtype=torch.float32, device=dev)
w_before = node["weights"].detach().clone()
obs = net.fit_bptt(inp, tgt, optimizer="sgd", lr=1e-3, update_steps=200, sampling_steps=100, verbose=False)
if dist.get_rank() == 0:
    print("losses:", [round(float(v), 6) for v in obs["loss"]], "max |dW|:", float((node["weights"].detach() - w_before).abs().max()))
dist.destroy_process_group()
