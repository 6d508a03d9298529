"""bench.py's extra.stepwise leg alone (the per-step / coupled / edge-memory paths at N = 4096 and 32768)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

print(json.dumps(bench.stepwise_extra(torch.device("cuda:0")), indent=1))
