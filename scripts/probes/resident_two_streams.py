"""Two host threads, each on its own stream -- or, with `same` as the second argument, both on ONE stream (the default stream is
every thread's unless it chooses another) --, issuing resident launches at the same time (different populations): results against a
launch per step, and no timeout.

    python scripts/probes/resident_two_streams.py [seconds] [same]
"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population, default_backend  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
shared = torch.cuda.Stream(device=dev) if len(sys.argv) > 2 and sys.argv[2] == "same" else None
errors = []


def make(n, seed):
    gen = torch.Generator().manual_seed(seed)
    u = lambda lo, hi: torch.rand(n, generator=gen) * (hi - lo) + lo
    return Population("li_tanh", torch.randn(n, n, generator=gen) * 1.3 / n ** 0.5,
                      dict(tau=u(0.5, 2.0), k=u(0.5, 1.5), eta=torch.randn(n, generator=gen) * 0.1), 1e-2, dev,
                      v0=torch.randn(n, generator=gen) * 0.5)


def reference(n, seed, T, calls, x):
    os.environ["RECTIPY_AMD_RESIDENT"] = "0"
    pop = make(n, seed)
    for _ in range(calls):
        pop.forward(T, x=x)
    torch.cuda.synchronize()
    return pop.v.clone()


def worker(n, seed, T, calls, x, want, t_end):
    stream = shared if shared is not None else torch.cuda.Stream(device=dev)
    rounds = 0
    with torch.cuda.stream(stream):
        while time.time() < t_end and not errors:
            pop = make(n, seed)
            for _ in range(calls):
                pop.forward(T, x=x)
            stream.synchronize()
            if not torch.equal(pop.v, want):
                errors.append(f"n={n}: mismatch in round {rounds}")
            rounds += 1
    print(f"n={n}: {rounds} rounds of {calls} resident calls x {T} steps", flush=True)


cfgs = [(512, 11, 60, 20), (260, 12, 45, 25)]
refs = []
for n, seed, T, calls in cfgs:
    x = (torch.randn(T, n, generator=torch.Generator().manual_seed(seed)) * 0.5).to(dev)
    refs.append((x, reference(n, seed, T, calls, x)))
os.environ["RECTIPY_AMD_RESIDENT"] = "1"
t_end = time.time() + budget
threads = [threading.Thread(target=worker, args=(n, seed, T, calls, x, want, t_end)) for (n, seed, T, calls), (x, want) in zip(cfgs, refs)]
for th in threads:
    th.start()
for th in threads:
    th.join()
torch.cuda.synchronize()
default_backend().async_status()
print("one stream, two threads:" if shared is not None else "two streams:", "FAILED " + "; ".join(errors) if errors else "ok, no timeout reported", flush=True)
sys.exit(1 if errors else 0)
