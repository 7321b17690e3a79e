"""Latency of the single-proof configurations through the C ABI:
config 1  range proof n = 64, m = 1, the reference's bench input (bench/Bench/RangeProof.hs:10-14)
config 2  aggregated range proof n = 64, m = 16 (nm = 1024)
config 5  arithmetic circuit, n = 2^12 gates, Q = 2 dense rows (Hadamard-style)
Each line: median wall time of the raw library call with pre-encoded inputs (host buffers in and
out, so PCIe staging and the host transcript are included), and a prove -> verify check."""
import argparse
import ctypes
import json
import os
import random
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bulletproofs_amd import Engine, _lib

R = _lib.R_ORDER
GEN = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
       0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def timed(fn, iters):
    fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3, min(ts) * 1e3


def range_config(eng, lib, name, n, vs, seed, iters, h):
    rng = random.Random(seed)
    m = len(vs)
    m2 = 1
    while m2 < m:
        m2 *= 2
    nm = n * m2
    k = nm.bit_length() - 1
    enc = _lib.enc_fr
    values, blinds = enc([v for v, _ in vs]), enc([b for _, b in vs])
    s_l = enc([rng.randrange(1, R) for _ in range(nm)])
    s_r = enc([rng.randrange(1, R) for _ in range(nm)])
    four = [enc([rng.randrange(1, R)]) for _ in range(4)]
    sc = ctypes.create_string_buffer(5 * 32)
    pts = ctypes.create_string_buffer((4 + 2 * k) * 64)

    def prove():
        rc = lib.bp_rangeproof_prove(eng.h, n, m, values, blinds, s_l, s_r, *four, 1, sc, pts)
        assert rc == 0, rc
    p_med, p_min = timed(prove, iters)
    vc = eng.add_two_mul_p([v for v, _ in vs], [GEN] * m, [b for _, b in vs], [h] * m)
    allpts = pts.raw + _lib.enc_points(vc + [None] * (m2 - m))
    out = ctypes.create_string_buffer(1)

    def verify():
        rc = lib.bp_rangeproof_verify_batch(eng.h, n, m2, 1, sc.raw, allpts, out)
        assert rc == 0, rc
    v_med, v_min = timed(verify, iters)
    assert out.raw[0] == 1, "engine proof did not verify"
    print(json.dumps(dict(bench=name, n_bits=n, m=m, nm=nm, ipp_rounds=k, prove_ms=round(p_med, 3),
                          prove_ms_min=round(p_min, 3), verify_ms=round(v_med, 3), verify_ms_min=round(v_min, 3),
                          verified=True)), flush=True)


def circuit_config(eng, lib, n, iters, h):
    rng = random.Random(5)
    enc = _lib.enc_fr
    a_l = [rng.randrange(R) for _ in range(n)]
    a_r = [rng.randrange(R) for _ in range(n)]
    a_o = [x * y % R for x, y in zip(a_l, a_r)]
    r, s = rng.randrange(R), rng.randrange(R)
    ones, zeros = [1] * n, [0] * n
    w_l, w_r, w_o = enc(ones + zeros), enc(zeros + ones), enc(zeros + zeros)
    w_v, cs = enc([1, 0, 0, 1]), enc([0, 0])
    vc = eng.add_two_mul_p([sum(a_l) % R, sum(a_r) % R], [GEN] * 2, [r, s], [h] * 2)
    args = [enc(a_l), enc(a_r), enc(a_o), enc([r, s]), enc([rng.randrange(1, R) for _ in range(3)]),
            enc([rng.randrange(1, R) for _ in range(n)]), enc([rng.randrange(1, R) for _ in range(n)]),
            enc([rng.randrange(1, R) for _ in range(5)])]
    npts = lib.bp_ac_proof_points(n)
    sc = ctypes.create_string_buffer(5 * 32)
    pts = ctypes.create_string_buffer(npts * 64)

    def prove():
        rc = lib.bp_ac_prove(eng.h, n, 2, 2, w_l, w_r, w_o, w_v, *args, sc, pts)
        assert rc == 0, rc
    p_med, p_min = timed(prove, iters)
    out = ctypes.create_string_buffer(1)
    vcb = _lib.enc_points(vc)

    def verify():
        rc = lib.bp_ac_verify(eng.h, n, 2, 2, w_l, w_r, w_o, w_v, cs, vcb, sc.raw, pts.raw, out)
        assert rc == 0, rc
    v_med, v_min = timed(verify, iters)
    assert out.raw[0] == 1, "engine circuit proof did not verify"
    print(json.dumps(dict(bench="config5_arith_circuit", gates=n, q=2, m=2, ipp_rounds=(npts - 10) // 2,
                          prove_ms=round(p_med, 3), prove_ms_min=round(p_min, 3), verify_ms=round(v_med, 3),
                          verify_ms_min=round(v_min, 3), verified=True)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--gates", type=int, default=4096)
    a = ap.parse_args()
    eng = Engine(0)
    lib = eng.lib
    _, _, h = eng.derive_generators(max(a.gates, 1024))
    range_config(eng, lib, "config1_rangeproof_64", 64, [(7238283, 827361)], 1, a.iters, h)
    rng = random.Random(2)
    vs = [(rng.randrange(0, 2**64), rng.randrange(1, R)) for _ in range(16)]
    range_config(eng, lib, "config2_rangeproof_64x16", 64, vs, 2, a.iters, h)
    circuit_config(eng, lib, a.gates, a.iters, h)


if __name__ == "__main__":
    main()
