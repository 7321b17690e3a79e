"""CPU tests: the C restatement (oracle/bp_ref.c, the timed CPU baseline) against the Python
oracle and the golden vectors."""
import ctypes
import os
import random
import subprocess

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from util import golden, ux, up, load_proof

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    so = os.path.join(ROOT, "oracle", "libbp_ref.so")
    src = os.path.join(ROOT, "oracle", "bp_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = ctypes.CDLL(so)
    L.bpref_init(0, 0, 64)
    return L


def test_generators(ref):
    g = ctypes.create_string_buffer(64 * 64)
    h = ctypes.create_string_buffer(64 * 64)
    h1 = ctypes.create_string_buffer(64)
    ref.bpref_get_generators(g, h, h1)
    assert _lib.dec_point(h1.raw) == o.h()
    assert _lib.dec_points(g.raw, 64) == o.gs(64)
    assert _lib.dec_points(h.raw, 64) == o.hs(64)


def test_point_mul_and_sum_exps(ref):
    rng = random.Random(21)
    out = ctypes.create_string_buffer(64)
    for e in golden()["kG"]:
        ref.bpref_point_mul(_lib.enc_point(o.GEN), _lib.enc_scalar(ux(e["k"])), out)
        assert _lib.dec_point(out.raw) == up(e["P"])
    for case in golden()["sumExps"]:
        es = [ux(s) for s in case["scalars"]]
        ps = [up(q) for q in case["points"]]
        ref.bpref_sum_exps(_lib.enc_scalars(es), _lib.enc_points(ps), ctypes.c_size_t(len(es)), out)
        assert _lib.dec_point(out.raw) == up(case["result"])
    p = o.point_mul(o.GEN, rng.randrange(1, o.R))
    ref.bpref_sum_exps(_lib.enc_scalars([5, o.R - 5]), _lib.enc_points([p, p]), ctypes.c_size_t(2), out)
    assert _lib.dec_point(out.raw) is None


def pack_one(pr, vcs, m2):
    ipp = pr["productProof"]
    sc = _lib.enc_scalars([pr["tBlinding"], pr["mu"], pr["t"], ipp["l"], ipp["r"]])
    rows = [pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"]] + ipp["lCommits"] + ipp["rCommits"]
    rows += [(vcs[j] if j < len(vcs) else None) for j in range(m2)]
    return sc, _lib.enc_points(rows)


def c_verify(ref, n, m2, pr, vcs):
    sc, pts = pack_one(pr, vcs, m2)
    out = ctypes.create_string_buffer(1)
    ref.bpref_verify_batch(n, m2, ctypes.c_size_t(1), sc, pts, ctypes.c_size_t(0), ctypes.c_size_t(1), out)
    return out.raw[0] != 0


def test_verify_golden_proofs(ref):
    g = golden()
    b = g["bench_proof"]
    pr = load_proof(b["proof"])
    vc = up(b["vCommit"])
    assert c_verify(ref, 64, 1, pr, [vc]) is True
    assert c_verify(ref, 64, 1, pr, [o.commit(b["v"] + 1, b["vBlinding"])]) is False
    assert c_verify(ref, 64, 1, dict(pr, t=(pr["t"] + 1) % o.R), [vc]) is False
    mp = g["multi_proof"]
    pr = load_proof(mp["proof"])
    vcs = [up(p) for p in mp["vCommits"]]
    assert c_verify(ref, 8, 4, pr, vcs) is True  # m = 3 padded to 4 with O
    ipp = dict(pr["productProof"])
    ipp["lCommits"], ipp["rCommits"] = ipp["rCommits"], ipp["lCommits"]
    assert c_verify(ref, 8, 4, dict(pr, productProof=ipp), vcs) is False


@pytest.mark.parametrize("n,m", [(1, 1), (4, 2), (8, 1)])
def test_verify_matches_python_oracle(ref, n, m):
    rng = random.Random(300 + 10 * n + m)
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, o.R)) for _ in range(m)]
    pr = o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m))
    vcs = [o.commit(v, b) for v, b in vs]
    assert c_verify(ref, n, m, pr, vcs) is True
    assert c_verify(ref, n, m, dict(pr, mu=(pr["mu"] + 1) % o.R), vcs) is False


def test_literal_affine_ladder_gives_the_same_points(ref):
    """bpref_set_mul_mode(1): the literal affine double-and-add of `mul` on a `PA` (one inversion per step) - the
    slower of the two CPU baselines bench.py reports - yields the same group elements as the Jacobian mode."""
    rng = random.Random(23)
    out = ctypes.create_string_buffer(64)
    try:
        ref.bpref_set_mul_mode(1)
        for e in golden()["kG"]:
            ref.bpref_point_mul(_lib.enc_point(o.GEN), _lib.enc_scalar(ux(e["k"])), out)
            assert _lib.dec_point(out.raw) == up(e["P"])
        ps = [o.point_mul(o.GEN, rng.randrange(1, o.R)) for _ in range(5)] + [None]
        es = [rng.randrange(o.R) for _ in range(4)] + [0, 7]
        ref.bpref_sum_exps(_lib.enc_scalars(es), _lib.enc_points(ps), ctypes.c_size_t(6), out)
        assert _lib.dec_point(out.raw) == o.sum_exps(es, ps)
        p = ps[0]
        ref.bpref_sum_exps(_lib.enc_scalars([5, o.R - 5, 2, 1]), _lib.enc_points([p, p, p, o.point_add(p, p)]),
                           ctypes.c_size_t(4), out)     # P - P, then 2P + 2P: the doubling branch of `<>`
        assert _lib.dec_point(out.raw) == o.point_mul(p, 4)
    finally:
        ref.bpref_set_mul_mode(0)
