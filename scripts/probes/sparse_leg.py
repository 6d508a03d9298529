"""bench.py's extra.sparse_p0.1_n32768 leg alone."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

print(json.dumps(bench.sparse_extra(torch.device("cuda:0")), indent=1))
