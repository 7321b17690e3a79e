"""GPU parity tests for bp_rangeproof_verify, the verifier that is total over its arguments like
MultiRangeProof.verifyProof (MultiRangeProof/Verifier.hs:19-88): any upper bound, any number of L/R pairs.
Restates the reference's remaining verifier tests on the GPU (test/Test/Protocol/RangeProof.hs:112-221):
prop_differentUpperBound, n = 128, and m = 6..10 (padding to 8 / 16 commitments with O).  Every verdict - and every
place where the reference raises instead of returning a Bool - is compared with the oracle."""
import os
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from test_gpu_verify import pack_batch

pytestmark = pytest.mark.gpu
R = o.R


@pytest.fixture(scope="module")
def eng(engine):
    engine.derive_generators(256)
    return engine


def both(eng, ub, vcs, pr):
    """-> (oracle outcome, engine outcome); 'raises' where the reference's evaluation fails."""
    try:
        exp = o.mrp_verify_proof(ub, vcs, pr)
    except (IndexError, ArithmeticError):
        exp = "raises"
    try:
        got = eng.verify_proof(ub, vcs, pr)
    except _lib.BpError as e:
        assert e.code == _lib.ERR_REFERENCE_FAILS, str(e)
        got = "raises"
    return exp, got


def honest(rng, n, m):
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
    m2 = 2 ** o.log2_ceil(m)
    pr = o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m2))
    return vs, [o.commit(v, b) for v, b in vs], pr


def forge_t(pr, n_old, n_new, m2):
    """The proof with t moved by delta(n_new) - delta(n_old): verifyTPoly then holds under the OTHER bound, so the
    reference goes on to evaluate verifyLRCommitment there (Verifier.hs:26-29 is a lazy `and`)."""
    y = o.shamir_y(pr["aCommit"], pr["sCommit"])
    z = o.shamir_z(pr["aCommit"], pr["sCommit"], y)
    t = (pr["t"] - o.delta(n_old, m2, y, z) + o.delta(n_new, m2, y, z)) % R
    return dict(pr, t=t)


BOUNDS = [2**8, 2**4, 2**16, 12, 1, 0, 2**7, 3, 2**64 - 1, 2**8 + 1, 2**200]


def test_different_upper_bound(eng):
    """prop_differentUpperBound (test/Test/Protocol/RangeProof.hs:173-181): a proof for 2^8 under other bounds,
    including bounds that are no power of two, n = 0, logBase2's Double rounding (2^64 - 1 -> 64) and n > 128."""
    rng = random.Random(301)
    vs, vcs, pr = honest(rng, 8, 1)
    seen = set()
    for ub in BOUNDS:
        exp, got = both(eng, ub, vcs, pr)
        assert got == exp, "upper bound %d: engine %r, oracle %r" % (ub, got, exp)
        seen.add(exp)
    assert both(eng, 2**8, vcs, pr) == (True, True)
    assert seen == {True, False}  # verifyTPoly fails first under every other bound


def test_different_upper_bound_past_tpoly(eng):
    """The same with t adjusted so that verifyTPoly HOLDS under the other bound: the reference then evaluates
    computeLRCommitment and IPP.verifyProof for that n - Map walk of mkOtherExponents for n no power of two,
    sumExps truncation, surplus L/R pairs - or fails at `challenges !! j` when the proof has too few pairs."""
    rng = random.Random(302)
    vs, vcs, pr = honest(rng, 8, 1)
    outcomes = {}
    for ub in BOUNDS:
        n_new = max(o.log_base2(ub), 0) if ub else 0
        f = forge_t(pr, 8, o.log_base2(ub), 1)
        y = o.shamir_y(f["aCommit"], f["sCommit"])
        z = o.shamir_z(f["aCommit"], f["sCommit"], y)
        x = o.shamir_x(f["aCommit"], f["sCommit"], f["t1Commit"], f["t2Commit"], y, z)
        assert o.verify_t_poly(o.log_base2(ub), vcs, f, x, y, z)
        exp, got = both(eng, ub, vcs, f)
        assert got == exp, "upper bound %d (n = %d): engine %r, oracle %r" % (ub, n_new, got, exp)
        outcomes[ub] = exp
    assert outcomes[2**8] is True
    assert outcomes[2**16] == "raises" and outcomes[2**200] == "raises" and outcomes[2**64 - 1] == "raises"
    assert outcomes[12] is False and outcomes[2**4] is False and outcomes[1] is False and outcomes[0] is False


def test_pair_count_differs(eng):
    """k_proof != log2(n m): a surplus pair is folded by mkChallenges and never indexed (a Bool); a missing pair makes
    the reference fail at `!!` - unless verifyTPoly already failed."""
    rng = random.Random(303)
    vs, vcs, pr = honest(rng, 8, 2)
    ipp = pr["productProof"]
    extra = o.point_mul(o.GEN, 77)
    longer = dict(pr, productProof=dict(ipp, lCommits=ipp["lCommits"] + [extra], rCommits=ipp["rCommits"] + [o.GEN]))
    shorter = dict(pr, productProof=dict(ipp, lCommits=ipp["lCommits"][:-1], rCommits=ipp["rCommits"][:-1]))
    uneven = dict(pr, productProof=dict(ipp, lCommits=ipp["lCommits"], rCommits=ipp["rCommits"][:-2]))  # zip truncates
    none = dict(pr, productProof=dict(ipp, lCommits=[], rCommits=[]))
    assert both(eng, 2**8, vcs, longer) == (False, False)
    assert both(eng, 2**8, vcs, shorter) == ("raises", "raises")
    assert both(eng, 2**8, vcs, uneven) == ("raises", "raises")
    assert both(eng, 2**8, vcs, none) == ("raises", "raises")
    bad_t = dict(shorter, t=(shorter["t"] + 1) % R)  # verifyTPoly is False: `and` never reaches the index error
    assert both(eng, 2**8, vcs, bad_t) == (False, False)
    assert both(eng, 2**8, [], pr) == ("raises", "raises")  # log2Ceil 0 = -1, 2 ^ (-1)


def test_n128_and_padded_m(eng):
    """n = 2^7 (m = 1, 2) and m = 6, 7, 9, 10 at n = 8 (O padding to 8 / 16 commitments): the reference's own test
    ranges (test/Test/Protocol/RangeProof.hs:112-145,196-208).  Proofs from the engine's prover, checked with the oracle's
    verifier; verdicts of bp_rangeproof_verify (honest, v + 1 commitment, a dropped commitment) against the oracle."""
    rng = random.Random(304)
    for n, m in [(128, 1), (128, 2), (8, 6), (8, 7), (8, 9), (8, 10)]:
        m2 = 2 ** o.log2_ceil(m)
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        bl = o.draw_blinders(rng, n * m2)
        pr = eng.prove(n, vs, *bl)
        vcs = [o.commit(v, b) for v, b in vs]
        assert len(pr["productProof"]["lCommits"]) == (n * m2).bit_length() - 1
        assert both(eng, 2**n, vcs, pr) == (True, True), (n, m)
        bad = [o.commit(vs[0][0] + 1, vs[0][1])] + vcs[1:]  # test_invalidCommitment (:183-194)
        assert both(eng, 2**n, bad, pr) == (False, False), (n, m)
        if m in (7, 10):  # one commitment fewer: another padding, same m2 for 7 -> 6, 10 -> 9
            assert both(eng, 2**n, vcs[:-1], pr) == (False, False), (n, m)
        # the batch entry on the same proof, commitments padded by the caller
        sc, pts, _ = pack_batch([pr, pr], [vcs, bad], m2)
        assert eng.verify_batch(n, m2, 2, sc, pts) == [True, False], (n, m)


def test_n128_proof_equals_oracle(eng):
    """The engine's n = 128 proof is the oracle's, byte for byte (MultiRangeProof/Prover.hs:47-123)."""
    rng = random.Random(305)
    vs = [(rng.randrange(0, 2**128), rng.randrange(1, R))]
    bl = o.draw_blinders(rng, 128)
    assert eng.prove(128, vs, *bl) == o.mrp_generate_proof(2**128, vs, *bl)


def test_general_path_equals_batch_engine(eng):
    """BP_VERIFY_GENERAL=1 sends honest-shape proofs down the general path too: same verdicts as the batch engine and
    the oracle, for honest and mutated proofs (n = 1 has zero IPP rounds; m = 3 pads with O)."""
    rng = random.Random(306)
    cases = []
    for n, m in [(1, 1), (2, 1), (8, 3), (16, 2)]:
        vs, vcs, pr = honest(rng, n, m)
        ipp = pr["productProof"]
        muts = [pr, dict(pr, mu=(pr["mu"] + 1) % R), dict(pr, productProof=dict(ipp, l=(ipp["l"] + 1) % R)),
                dict(pr, sCommit=o.point_add(pr["sCommit"], o.GEN))]
        if ipp["lCommits"]:
            muts.append(dict(pr, productProof=dict(ipp, lCommits=ipp["rCommits"], rCommits=ipp["lCommits"])))
        cases += [(2**n, vcs, p) for p in muts]
    fast = [both(eng, *c) for c in cases]
    os.environ["BP_VERIFY_GENERAL"] = "1"
    try:
        general = [both(eng, *c) for c in cases]
    finally:
        del os.environ["BP_VERIFY_GENERAL"]
    assert fast == general
    assert all(e == g for e, g in fast)
    assert {e for e, _ in fast} == {True, False}


def test_n256_general_path_accepts(eng):
    """n = 256 is beyond the batch engine (n <= 128) and beyond generateProof's `upperBound < _r` check, but
    verifyProof takes it: an oracle proof from generateProofUnsafe verifies on the general path, its mutations do not."""
    rng = random.Random(307)
    n = 256
    vs = [(rng.randrange(0, R), rng.randrange(1, R))]
    pr = o.mrp_generate_proof_unsafe(2**n, vs, *o.draw_blinders(rng, n))
    vcs = [o.commit(*vs[0])]
    assert eng.verify_proof(2**n, vcs, pr) is True
    assert o.mrp_verify_proof(2**n, vcs, pr) is True
    assert eng.verify_proof(2**n, vcs, dict(pr, mu=(pr["mu"] + 1) % R)) is False
    assert eng.verify_proof(2**n, [o.commit(vs[0][0] + 1, vs[0][1])], pr) is False


def test_invalid_encodings_do_not_verify(eng):
    """Values the reference's types cannot hold (off-curve point, coordinate >= p, scalar >= r): verdict False on both
    paths, as in the batch engine's k_vcheck."""
    rng = random.Random(308)
    vs, vcs, pr = honest(rng, 8, 1)
    off = (pr["aCommit"][0], (pr["aCommit"][1] + 1) % o.P)
    for ub in (2**8, 12):
        f = forge_t(pr, 8, o.log_base2(ub), 1) if ub != 2**8 else pr
        assert eng.verify_proof(ub, vcs, dict(f, t1Commit=off)) is False
        assert eng.verify_proof(ub, [off], f) is False
        assert eng.verify_proof(ub, vcs, dict(f, mu=R + 5)) is False
