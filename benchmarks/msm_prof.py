"""Kernel-level look at the MSM at N = 2^logn: plain bp_msm_dev and prepared points (run under rocprofv3)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from bulletproofs_amd import Engine  # noqa: E402
import bench  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
n = 1 << logn
eng = Engine(0)
ks, pts = bench.make_points(eng, n, 3)
es = bench.make_scalars(sys.argv[3] if len(sys.argv) > 3 else "uniform", n, 5)
d_p = torch.from_numpy(pts).cuda()
h_s = torch.from_numpy(es).pin_memory()
d_s = h_s.cuda()
d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
if mode in ("both", "plain"):
    for _ in range(4):
        eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr())
    eng.sync()
if mode in ("both", "prepared"):
    mp = eng.msm_points(d_points=d_p.data_ptr(), n=n)
    for _ in range(4):
        mp.run_dev(d_s.data_ptr(), d_o.data_ptr())
    eng.sync()
    for _ in range(4):
        mp.run(h_s.data_ptr())
    mp.close()
eng.close()
