"""GPU parity tests for the Pippenger MSM (`sumExps`, Utils.hs:76-80) through the
C ABI: bit-exact affine results against the oracle on small seeded inputs, the
golden vectors, adversarial bucket shapes, and size-independent properties
(linearity, known-dlog closed form) at BASELINE.json's full sizes."""
import ctypes
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from util import golden, ux, up

pytestmark = pytest.mark.gpu
R = o.R


def test_golden_sum_exps(engine):
    for case in golden()["sumExps"]:
        es = [ux(s) for s in case["scalars"]]
        ps = [up(q) for q in case["points"]]
        assert engine.sum_exps(es, ps) == up(case["result"])


def test_truncation_and_empty(engine):
    """Utils.hs:79-80."""
    p = o.gs(3)
    assert engine.sum_exps([], p) is None
    assert engine.sum_exps([5], []) is None
    assert engine.sum_exps([5, 7, 9], p[:2]) == o.sum_exps([5, 7], p[:2])


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 200, 1000])
def test_small_vs_oracle(engine, n):
    rng = random.Random(50 + n)
    base = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(min(n, 24))]
    ps = [rng.choice(base + [None]) for _ in range(n)]
    es = [rng.choice([0, 1, R - 1, rng.randrange(R), rng.randrange(R), rng.randrange(2**64)]) for _ in range(n)]
    # closed form through the known discrete logs keeps the oracle side fast
    assert engine.sum_exps(es, ps) == o.sum_exps(es, ps)


def _msm_window(engine, es, ps, c):
    """bp_msm_dev with a forced window size, staging through torch tensors."""
    import torch
    n = len(es)
    s = torch.frombuffer(bytearray(_lib.enc_scalars(es)), dtype=torch.uint8).cuda()
    p = torch.frombuffer(bytearray(_lib.enc_points(ps)), dtype=torch.uint8).cuda()
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    engine.msm_dev(s.data_ptr(), p.data_ptr(), n, out.data_ptr(), window_bits=c)
    engine.sync()
    return _lib.dec_point(bytes(out.cpu().numpy()))


@pytest.mark.parametrize("c", [1, 2, 5, 8, 11, 14, 16])
def test_every_window_size(engine, c):
    rng = random.Random(60 + c)
    n = 96
    base = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(16)]
    ps = [rng.choice(base) for _ in range(n)]
    es = [rng.choice([0, 1, 2, R - 1, R - 2, 2**255, rng.randrange(R), rng.randrange(R)]) for _ in range(n)]
    assert _msm_window(engine, es, ps, c) == o.sum_exps(es, ps)


def test_adversarial_buckets(engine):
    """All terms in one bucket (heavy path), P + P, P + (-P), identity points."""
    rng = random.Random(70)
    p = o.point_mul(o.GEN, rng.randrange(1, R))
    q = o.point_mul(o.GEN, rng.randrange(1, R))
    n = 5000
    # same point, same scalar: every window has a single bucket of n entries
    assert engine.sum_exps([3] * n, [p] * n) == o.point_mul(p, 3 * n)
    # {0,1} scalars (the shape commitBitVectors sees, RangeProof/Internal.hs:136)
    bits = [rng.randrange(2) for _ in range(n)]
    pts = [p if i % 2 else q for i in range(n)]
    expect = o.point_add(o.point_mul(p, sum(b for i, b in enumerate(bits) if i % 2)),
                         o.point_mul(q, sum(b for i, b in enumerate(bits) if not i % 2)))
    assert engine.sum_exps(bits, pts) == expect
    # exact cancellation
    assert engine.sum_exps([7, R - 7] * 1000, [p] * 2000) is None
    # identity points carry any scalar
    assert engine.sum_exps([rng.randrange(R) for _ in range(100)], [None] * 100) is None
    # -1 scalars (aR = aL - 1, RangeProof/Internal.hs:87-88)
    assert engine.sum_exps([R - 1] * 300, [p] * 300) == o.point_mul(p, (R - 300) % R)


def _known_dlog_points(n, seed):
    """n points k_i*G with known k_i, built on the GPU itself (bp_point_mul is
    pinned against the oracle above); returns (ks, encoded points)."""
    rng = random.Random(seed)
    ks = [rng.randrange(1, R) for _ in range(n)]
    return ks


@pytest.mark.parametrize("logn", [16, 20])
def test_full_size_known_dlog(engine, logn):
    """BASELINE.json config 3 sizes: sum e_i*(k_i*G) must equal (sum e_i k_i mod r)*G."""
    n = 1 << logn
    rng = random.Random(80 + logn)
    ks = [rng.randrange(1, R) for _ in range(n)]
    pts_raw = engine.point_mul_raw(_lib.enc_scalars(ks), _lib.enc_point(o.GEN) * n, n)
    es = [rng.randrange(R) for _ in range(n)]
    got = _lib.dec_point(engine.msm_raw(_lib.enc_scalars(es), pts_raw, n))
    expect = o.point_mul(o.GEN, sum(e * k for e, k in zip(es, ks)) % R)
    assert got == expect
    # linearity: MSM(a*e) == a * MSM(e)
    a = rng.randrange(2, R)
    got2 = _lib.dec_point(engine.msm_raw(_lib.enc_scalars([a * e % R for e in es]), pts_raw, n))
    assert got2 == o.point_mul(expect, a)


# ---- bp_msm_points: a fixed point list prepared once (shifted copies, no window doublings) --------------------------
@pytest.mark.parametrize("n", [1, 7, 300, 5000])
def test_prepared_points_vs_oracle(engine, n):
    """Same group element as sumExps, byte for byte, for several scalar vectors on one handle; O among the points,
    scalars 0, 1, r - 1, values >= r (they act modulo r)."""
    import torch
    rng = random.Random(150 + n)
    base = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(min(n, 12))]
    ps = [rng.choice(base + [None]) for _ in range(n)]
    mp = engine.msm_points(points=_lib.enc_points(ps))
    d_out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        es = [rng.choice([0, 1, R - 1, rng.randrange(R), rng.randrange(R), rng.randrange(2**64), R + 5])
              for _ in range(n)]
        exp = o.sum_exps([e % R for e in es], ps)
        assert _lib.dec_point(mp.run(_lib.enc_scalars(es))) == exp
        d_s = torch.frombuffer(bytearray(_lib.enc_scalars(es)), dtype=torch.uint8).cuda()
        mp.run_dev(d_s.data_ptr(), d_out.data_ptr())
        engine.sync()
        assert _lib.dec_point(bytes(d_out.cpu().numpy())) == exp
    mp.close()


@pytest.mark.parametrize("kind", ["uniform", "64bit", "01", "one_bucket"])
def test_prepared_points_full_size_structured_scalars(engine, kind):
    """2^18 points with known discrete logs: the result must be (sum e_i k_i mod r) * G for the scalar shapes of
    BASELINE config 3 (uniform, 64-bit, {0,1}: the last two pile onto bucket 0, the wave-aggregated atomics and the
    heavy-bucket path), through the chunked pinned-memory upload and through bp_msm_dev on the same inputs."""
    import numpy as np
    import torch
    n = 1 << 18
    rng = np.random.default_rng(160)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x7F
    g = np.frombuffer(_lib.enc_point(o.GEN), dtype=np.uint8)
    pts = np.frombuffer(engine.point_mul_raw(ks.tobytes(), np.tile(g, n).tobytes(), n), dtype=np.uint8).copy()
    es = np.zeros((n, 32), dtype=np.uint8)
    if kind == "uniform":
        es[:] = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        es[:, 31] &= 0x7F
    elif kind == "64bit":
        es[:, :8] = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    elif kind == "01":
        es[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)
    else:
        es[:, 2] = 3  # every scalar 3 * 2^16: one bucket of window 1 holds all entries
    k_int = [int.from_bytes(bytes(r_), "little") for r_ in ks]
    e_int = [int.from_bytes(bytes(r_), "little") for r_ in es]
    exp = o.point_mul(o.GEN, sum(a * b for a, b in zip(e_int, k_int)) % R)
    d_p = torch.from_numpy(pts).cuda()
    mp = engine.msm_points(d_points=d_p.data_ptr(), n=n)
    h_s = torch.from_numpy(es).pin_memory()
    assert _lib.dec_point(mp.run(h_s.data_ptr())) == exp           # pinned host scalars: chunked upload
    assert _lib.dec_point(mp.run(es.tobytes())) == exp             # pageable host scalars
    d_s = h_s.cuda()
    d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
    mp.run_dev(d_s.data_ptr(), d_o.data_ptr())
    engine.sync()
    assert _lib.dec_point(bytes(d_o.cpu().numpy())) == exp
    engine.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, d_o.data_ptr())   # the plain bucket MSM on the same inputs
    engine.sync()
    assert _lib.dec_point(bytes(d_o.cpu().numpy())) == exp
    mp.close()


# ---- bp_msm_dev_hs: scalars from host memory, uploaded in pieces behind which the bucket work runs -------------------
@pytest.mark.parametrize("kind", ["uniform", "64bit", "01", "minus1"])
def test_host_scalar_msm_in_pieces(engine, kind):
    """2^19 + 777 points with known discrete logs, in one piece (the default) and in two / three uneven pieces
    (BP_MSM_HS_PARTS, each piece with its own bucket array): the result must be (sum e_i k_i mod r) * G and byte-equal to
    bp_msm_dev's, from pinned and from pageable scalars, call after call on one context (the staging buffer is reused);
    small sums and n = 0 as well.  `minus1`: scalars in {0, r - 1},
    the shape of aR in commitBitVectors (RangeProof/Internal.hs:136-147)."""
    import numpy as np
    import torch
    n = (1 << 19) + 777
    rng = np.random.default_rng(170)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x7F
    g = np.frombuffer(_lib.enc_point(o.GEN), dtype=np.uint8)
    pts = np.frombuffer(engine.point_mul_raw(ks.tobytes(), np.tile(g, n).tobytes(), n), dtype=np.uint8).copy()
    es = np.zeros((n, 32), dtype=np.uint8)
    if kind == "uniform":
        es[:] = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        es[:, 31] &= 0x7F
    elif kind == "64bit":
        es[:, :8] = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    elif kind == "01":
        es[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)
    else:
        rm1 = np.frombuffer(_lib.enc_scalar(R - 1), dtype=np.uint8)
        es[rng.integers(0, 2, size=n).astype(bool)] = rm1
    k_int = [int.from_bytes(bytes(r_), "little") for r_ in ks]
    e_int = [int.from_bytes(bytes(r_), "little") for r_ in es]
    d_p = torch.from_numpy(pts).cuda()
    h_s = torch.from_numpy(es).pin_memory()
    d_s = h_s.cuda()
    d_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
    d_r = torch.zeros(64, dtype=torch.uint8, device="cuda")
    import os
    try:
        for parts, m in ((1, n), (2, n), (2, 5000), (3, n), (1, 5000), (2, 1), (1, 0), (2, n)):
            os.environ["BP_MSM_HS_PARTS"] = str(parts)
            exp = o.point_mul(o.GEN, sum(a * b for a, b in zip(e_int[:m], k_int[:m])) % R)
            engine.msm_dev(d_s.data_ptr(), d_p.data_ptr(), m, d_r.data_ptr())
            engine.msm_dev_hs(h_s.data_ptr(), d_p.data_ptr(), m, d_o.data_ptr())
            engine.sync()
            assert _lib.dec_point(bytes(d_o.cpu().numpy())) == exp, (kind, parts, m)
            assert torch.equal(d_o, d_r)
        pageable = es.copy()
        engine.msm_dev_hs(pageable.ctypes.data, d_p.data_ptr(), n, d_o.data_ptr())
        engine.sync()
        assert torch.equal(d_o, d_r)
    finally:
        del os.environ["BP_MSM_HS_PARTS"]
