"""bp_msm_dev at N = 2^logn with forced window sizes (resident scalars)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from bulletproofs_amd import Engine  # noqa: E402
import bench  # noqa: E402

eng = Engine(0)
for logn in (17, 18, 19, 20):
    n = 1 << logn
    ks, pts = bench.make_points(eng, n, 3)
    es = bench.make_scalars("uniform", n, 5)
    d_p = torch.from_numpy(pts).cuda()
    d_s = torch.from_numpy(es).cuda()
    d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
    res = {}
    ref = None
    for c in (0, 11, 12, 13, 14, 15, 16):
        def run():
            eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr(), c)
            eng.sync()
        run()
        out = bytes(d_o.cpu().numpy())
        ref = ref or out
        assert out == ref
        t0 = time.perf_counter()
        for _ in range(8):
            run()
        res[c] = round((time.perf_counter() - t0) / 8 * 1e3, 3)
    print(logn, res, flush=True)
