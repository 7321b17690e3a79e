"""MSM throughput probe (BASELINE.json config 3): points/s for N = 2^logn."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bulletproofs_amd import Engine, _lib


def make_inputs(eng, n, seed):
    rng = np.random.default_rng(seed)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x7F  # < 2^255 < r
    g = np.frombuffer(_lib.enc_point((_lib_gen())), dtype=np.uint8)
    pts = eng.point_mul_raw(ks.tobytes(), np.tile(g, n).tobytes(), n)
    es = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    es[:, 31] &= 0x7F
    return es, np.frombuffer(pts, dtype=np.uint8)


def _lib_gen():
    return (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
            0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, nargs="+", default=[16, 18, 20])
    ap.add_argument("--c", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    eng = Engine(0)
    for logn in a.logn:
        n = 1 << logn
        es, pts = make_inputs(eng, n, logn)
        d_s = torch.from_numpy(es.copy()).cuda()
        d_p = torch.from_numpy(pts.copy()).cuda()
        d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr(), 0)
        eng.sync()
        ref = bytes(d_o.cpu().numpy())
        for _ in range(2):
            eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr(), a.c)
        eng.sync()
        if bytes(d_o.cpu().numpy()) != ref:
            raise SystemExit("window_bits = %d gives another point than the default plan at 2^%d" % (a.c, logn))
        t0 = time.perf_counter()
        for _ in range(a.iters):
            eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr(), a.c)
        eng.sync()
        dt = (time.perf_counter() - t0) / a.iters
        print(json.dumps(dict(bench="msm", logn=logn, c=a.c, ms=dt * 1e3, Mpoints_per_s=n / dt / 1e6)), flush=True)


if __name__ == "__main__":
    main()
