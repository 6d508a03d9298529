"""Read+write streaming ceiling of the chip, for kernels that rewrite a matrix in place (the RLS pass: 8 n^2 bytes per
update): device-to-device copy of 4 GiB (hipMemcpyAsync via torch) and an in-place scale (read + write of the same
lines), bytes moved / time."""
import json

import torch

dev = torch.device("cuda:0")
n = 32768
a = torch.randn(n, n, device=dev)
b = torch.empty_like(a)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


res = {"copy_GBps": 2 * a.numel() * 4 / timed(lambda: b.copy_(a)) / 1e6,
       "inplace_scale_GBps": 2 * a.numel() * 4 / timed(lambda: a.mul_(1.0000001)) / 1e6,
       "read_only_sum_GBps": a.numel() * 4 / timed(lambda: a.sum()) / 1e6}
print(json.dumps({k: round(v, 1) for k, v in res.items()}), flush=True)
