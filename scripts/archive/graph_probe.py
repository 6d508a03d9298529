"""Would a hipGraph help the launch-bound small-network loop?  Capture T step launches (rp_forward issues them on
torch's current stream, so torch.cuda.graph records them) and compare replay time per step with direct issue."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
for N in (512, 1024, 2048, 4096):
    W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    T = 200
    x = torch.zeros(T, 1, device=dev)
    pop.forward(T, x=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pop.forward(T, x=x)
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / (20 * T) * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pop.forward(T, x=x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            pop.forward(T, x=x)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (20 * T) * 1e6
    print(json.dumps(dict(N=N, direct_us_per_step=round(direct, 2), graph_replay_us_per_step=round(graph, 2),
                          finite=bool(torch.isfinite(pop.v).all()))), flush=True)
