"""GPU parity tests for the prover path: bp_ipp_prove (InnerProductProof/Prover.hs:17-92),
bp_rangeproof_prove (MultiRangeProof/Prover.hs:19-123, RangeProof/Prover.hs:15-32) and
bp_upper_bound_bits, bit for bit against the oracle on the same inputs and blinders, plus
prove -> verify round trips through the engine's own batch verifier."""
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from util import golden, up, load_proof, load_blinders
from test_gpu_verify import pack_batch

pytestmark = pytest.mark.gpu
R = o.R


@pytest.fixture(scope="module")
def eng_gens(engine):
    engine.set_generators(o.gs(512), o.hs(512), o.h())
    return engine


def rand_point(rng):
    return o.point_mul(o.GEN, rng.randrange(1, R))


@pytest.mark.parametrize("n", [0, 1, 2, 8, 32])
def test_ipp_prove_vs_oracle(engine, n):
    """generateProof on an arbitrary base, as test/Test/Protocol/InnerProductProof builds it."""
    rng = random.Random(4100 + n)
    b_gs = [rand_point(rng) for _ in range(n)]
    b_hs = [rand_point(rng) for _ in range(n)]
    b_h = rand_point(rng)
    ls = [rng.randrange(R) for _ in range(n)]
    rs = [rng.randrange(R) for _ in range(n)]
    if n >= 2:
        ls[1] = 0                       # zero scalars and the identity must pass through
        b_gs[0] = o.O if n >= 8 else b_gs[0]
    c = o.dot(ls, rs) if n else 0
    P = o.point_add(o.point_add(o.sum_exps(ls, b_gs), o.sum_exps(rs, b_hs)), o.point_mul(b_h, c))
    exp = o.ipp_generate_proof(b_gs, b_hs, b_h, P, ls, rs)
    got = engine.ipp_prove(b_gs, b_hs, b_h, P, ls, rs)
    assert got == exp
    if n:
        assert o.ipp_verify_proof(n, b_gs, b_hs, b_h, P, got)


def test_ipp_prove_rejects_non_power_of_two(engine):
    with pytest.raises(_lib.BpError):
        engine.ipp_prove([o.GEN] * 3, [o.GEN] * 3, o.GEN, o.GEN, [1, 2, 3], [4, 5, 6])


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (8, 1), (4, 2), (8, 3), (16, 4), (32, 5), (64, 1)])
def test_rangeproof_prove_vs_oracle(eng_gens, n, m):
    rng = random.Random(5200 + 10 * n + m)
    m2 = 2 ** o.log2_ceil(m)
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
    if m > 1:
        vs[0] = (2**n - 1, vs[0][1])   # boundary values
        vs[1] = (0, vs[1][1])
    blinders = o.draw_blinders(rng, n * m2)
    exp = o.mrp_generate_proof(2**n, vs, *blinders)
    got = eng_gens.prove(n, vs, *blinders)
    assert got == exp
    # and the engine's verifier accepts it
    vcs = [[o.commit(v, b) for v, b in vs]]
    sc, pts, _ = pack_batch([got], vcs, m2)
    assert eng_gens.verify_batch(n, m2, 1, sc, pts) == [True]


def test_bench_proof_golden(eng_gens):
    """The reference's bench input (bench/Bench/RangeProof.hs:10-14), n = 64, with the committed blinders."""
    b = golden()["bench_proof"]
    got = eng_gens.prove(64, [(b["v"], b["vBlinding"])], *load_blinders(b["blinders"]))
    assert got == load_proof(b["proof"])


def test_multi_proof_golden(eng_gens):
    """m = 3 (padded to 4), n = 8 fixture with its committed blinders."""
    b = golden()["multi_proof"]
    vs = [(int(v, 16), int(bl, 16)) for v, bl in b["vs"]]
    got = eng_gens.prove(8, vs, *load_blinders(b["blinders"]))
    assert got == load_proof(b["proof"])


def test_errors_mirror_reference(eng_gens):
    """RangeProofError constructors (RangeProof/Internal.hs:40-45; MultiRangeProof/Prover.hs:25-38)."""
    rng = random.Random(9)
    n = 8
    bl = o.draw_blinders(rng, n)
    with pytest.raises(_lib.BpError) as ei:
        eng_gens.prove(n, [(2**n, 5)], *bl)                 # ValuesNotInRange
    assert ei.value.code == -5
    with pytest.raises(o.RangeProofError):
        o.mrp_generate_proof(2**n, [(2**n, 5)], *bl)
    assert _lib.upper_bound_bits(o.R) == -6                  # UpperBoundTooLarge
    assert _lib.upper_bound_bits(2**300) == -6
    assert _lib.upper_bound_bits(2**64 + 1) == -7            # NNotPowerOf2: upper bound not 2^n
    assert _lib.upper_bound_bits(2**24) == -7                # n = 24 not a power of two
    assert _lib.upper_bound_bits(2**64) == 64
    assert _lib.upper_bound_bits(2**1) == 1


@pytest.mark.parametrize("m,bad", [(1, 0), (2, 0), (2, 1)])
def test_unsafe_out_of_range_matches_reference_quirk(eng_gens, m, bad):
    """generateProofUnsafe on an out-of-range value: the digits beyond n are kept by fillWithZeros
    (RangeProof/Internal.hs:93-96), enter aCommit and shift later values, and the proof does not verify.
    The reference's own suite only covers the checked path (prop_valueNotInRange,
    test/Test/Protocol/RangeProof.hs:147-157); this case is pinned by the oracle's restatement alone."""
    rng = random.Random(600 + 2 * m + bad)
    n = 8
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
    vs[bad] = (2**n + rng.randrange(0, 2**n), vs[bad][1])
    bl = o.draw_blinders(rng, n * m)
    exp = o.mrp_generate_proof_unsafe(2**n, vs, *bl)
    got = eng_gens.prove(n, vs, *bl, checked=False)
    assert got == exp
    vcs = [[o.commit(v, b) for v, b in vs]]
    sc, pts, _ = pack_batch([got], vcs, m)
    assert eng_gens.verify_batch(n, m, 1, sc, pts) == [o.mrp_verify_proof(2**n, vcs[0], exp)] == [False]


def test_prove_many_verify_batch(eng_gens):
    """Round trip at batch scale: 24 engine-made proofs for (n, m) = (16, 2), a third corrupted."""
    rng = random.Random(31)
    n, m = 16, 2
    proofs, vcs, exp = [], [], []
    for i in range(24):
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        pr = eng_gens.prove(n, vs, *o.draw_blinders(rng, n * m))
        vc = [o.commit(v, b) for v, b in vs]
        if i % 3 == 2:
            pr["t"] = (pr["t"] + 1) % R
        proofs.append(pr)
        vcs.append(vc)
        exp.append(i % 3 != 2)
    sc, pts, _ = pack_batch(proofs, vcs, m)
    assert eng_gens.verify_batch(n, m, len(proofs), sc, pts) == exp


# ---- IPP.verifyProof on a caller-supplied base (InnerProductProof/Verifier.hs:18-73) -------------------------------
def _ipp_instance(engine, rng, n):
    ks = [rng.randrange(1, R) for _ in range(2 * n + 1)]
    pts = engine.mul([o.GEN] * (2 * n + 1), ks)
    b_gs, b_hs, b_h = pts[:n], pts[n:2 * n], pts[2 * n]
    ls = [rng.randrange(R) for _ in range(n)]
    rs = [rng.randrange(R) for _ in range(n)]
    P = engine.sum_exps(ls + rs + [o.dot(ls, rs) if n else 0], b_gs + b_hs + [b_h])
    return b_gs, b_hs, b_h, ls, rs, P


@pytest.mark.parametrize("n", [1, 2, 8, 32])
def test_ipp_verify_vs_oracle(engine, n):
    """Honest proofs from the ORACLE's prover, then mutated l, swapped L/R, a wrong commitment and a wrong base:
    the verdict equals the oracle's ipp_verify_proof each time."""
    rng = random.Random(4300 + n)
    b_gs, b_hs, b_h, ls, rs, P = _ipp_instance(engine, rng, n)
    proof = o.ipp_generate_proof(b_gs, b_hs, b_h, P, ls, rs)
    cases = [("honest", b_gs, b_hs, b_h, P, proof),
             ("l+1", b_gs, b_hs, b_h, P, dict(proof, l=(proof["l"] + 1) % R)),
             ("P+G", b_gs, b_hs, b_h, o.point_add(P, o.GEN), proof),
             ("wrong bH", b_gs, b_hs, o.GEN, P, proof)]
    if n >= 2:
        cases.append(("swap L/R", b_gs, b_hs, b_h, P, dict(proof, lCommits=proof["rCommits"], rCommits=proof["lCommits"])))
        cases.append(("Gs rotated", b_gs[1:] + b_gs[:1], b_hs, b_h, P, proof))
        cases.append(("L_0 = O", b_gs, b_hs, b_h, P, dict(proof, lCommits=[o.O] + proof["lCommits"][1:])))
    for name, g, h, u, p, pr in cases:
        exp = o.ipp_verify_proof(n, g, h, u, p, pr)
        assert engine.ipp_verify(n, g, h, u, p, pr) == exp, name
        assert exp == (name == "honest")


def test_ipp_verify_1024_generators(engine):
    """n = 1024 (the size of config 2's inner-product argument) on a random caller-supplied base: proof from
    bp_ipp_prove (itself oracle-checked at smaller n), verdicts of the honest and a mutated proof against the oracle."""
    rng = random.Random(4400)
    n = 1024
    b_gs, b_hs, b_h, ls, rs, P = _ipp_instance(engine, rng, n)
    proof = engine.ipp_prove(b_gs, b_hs, b_h, P, ls, rs)
    assert len(proof["lCommits"]) == 10
    assert engine.ipp_verify(n, b_gs, b_hs, b_h, P, proof) is True
    assert o.ipp_verify_proof(n, b_gs, b_hs, b_h, P, proof) is True
    bad = dict(proof, r=(proof["r"] + 1) % R)
    assert engine.ipp_verify(n, b_gs, b_hs, b_h, P, bad) is False
    bad2 = dict(proof, rCommits=proof["rCommits"][:5] + [o.GEN] + proof["rCommits"][6:])
    assert engine.ipp_verify(n, b_gs, b_hs, b_h, P, bad2) is False


def test_ipp_verify_reference_edge_semantics(engine):
    """What the reference's lists do at the edges: n = 0 (one exponent, empty sums), n not a power of two (the Map walk
    of mkOtherExponents leaves keys out), bases shorter than n (`sumExps` truncates, Utils.hs:79), more rounds than
    logBase2 n, and fewer (the reference's `!!` fails: BP_ERR_ARG)."""
    rng = random.Random(4500)
    pts = engine.mul([o.GEN] * 20, [rng.randrange(1, R) for _ in range(20)])
    b_gs, b_hs, b_h = pts[:8], pts[8:16], pts[16]
    empty = dict(lCommits=[], rCommits=[], l=0, r=0)
    for P in (o.O, o.GEN):
        assert engine.ipp_verify(0, [], [], b_h, P, empty) == o.ipp_verify_proof(0, [], [], b_h, P, empty) == (P is o.O)
    # arbitrary (dishonest) proofs: only the verdict's agreement with the oracle matters
    for n, ng, nh, k in [(3, 8, 8, 1), (6, 8, 8, 2), (5, 3, 2, 3), (8, 5, 8, 3), (8, 8, 8, 4), (4, 8, 8, 3), (1, 1, 1, 2)]:
        pr = dict(lCommits=pts[:k], rCommits=pts[3:3 + k], l=rng.randrange(R), r=rng.randrange(R))
        exp = o.ipp_verify_proof(n, b_gs[:ng], b_hs[:nh], b_h, pts[17], pr)
        assert engine.ipp_verify(n, b_gs[:ng], b_hs[:nh], b_h, pts[17], pr) == exp
    # positive cases at the edge: n = 1 without rounds, where cProof = l*G_0 + r*H_0 + (l r)*bH has no challenge in it;
    # empty base lists drop their term (sumExps [s] [] = O)
    l1, r1 = rng.randrange(1, R), rng.randrange(1, R)
    for ng, nh in [(1, 1), (0, 1), (1, 0), (0, 0)]:
        P = o.point_mul(b_h, l1 * r1 % R)
        if ng:
            P = o.point_add(P, o.point_mul(b_gs[0], l1))
        if nh:
            P = o.point_add(P, o.point_mul(b_hs[0], r1))
        pr = dict(lCommits=[], rCommits=[], l=l1, r=r1)
        assert o.ipp_verify_proof(1, b_gs[:ng], b_hs[:nh], b_h, P, pr) is True
        assert engine.ipp_verify(1, b_gs[:ng], b_hs[:nh], b_h, P, pr) is True
        assert engine.ipp_verify(1, b_gs[:ng], b_hs[:nh], b_h, o.point_add(P, o.GEN), pr) is False
    # fewer rounds than logBase2 n: `challenges !! j` fails in the reference
    with pytest.raises(_lib.BpError):
        engine.ipp_verify(8, b_gs, b_hs, b_h, pts[17], dict(lCommits=pts[:2], rCommits=pts[2:4], l=1, r=1))
