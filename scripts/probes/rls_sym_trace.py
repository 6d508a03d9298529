import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd.rls import RLS
n, T, k = 32768, 10, 3
dev = torch.device("cuda:0")
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
for rows in (64,):
    edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
    edge.symmetric = True
    edge.SYMMETRIC_ROWS = rows
    edge.sweep(X, Y, 1)
    torch.cuda.synchronize()
    del edge
