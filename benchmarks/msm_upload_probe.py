"""Timing of bp_msm_points_run (host scalars, chunked upload) against run_dev and plain bp_msm_dev at N = 2^20."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from bulletproofs_amd import Engine  # noqa: E402
import bench  # noqa: E402

n = 1 << 20
eng = Engine(0)
ks, pts = bench.make_points(eng, n, 3)
es = bench.make_scalars("uniform", n, 5)
d_p = torch.from_numpy(pts).cuda()
h_s = torch.from_numpy(es).pin_memory()
d_s = h_s.cuda()
d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
mp = eng.msm_points(d_points=d_p.data_ptr(), n=n)


def t(fn, it=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    return (time.perf_counter() - t0) / it * 1e3


def dev():
    mp.run_dev(d_s.data_ptr(), d_o.data_ptr())
    eng.sync()


def plain():
    eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr())
    eng.sync()


def copy_only():
    d_s.copy_(h_s, non_blocking=True)
    torch.cuda.current_stream().synchronize()


print("chunks", os.environ.get("BP_MSM_UPLOAD_CHUNKS"), "plain_dev %.3f prepared_dev %.3f prepared_host %.3f torch_h2d %.3f" %
      (t(plain), t(dev), t(lambda: mp.run(h_s.data_ptr())), t(copy_only)))
