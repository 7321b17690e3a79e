"""GPU parity tests for the batch verifier state machine (bp_vbatch_*), with the
Fiat-Shamir hashing done by the oracle's transcript functions: every
intermediate the reference exposes (computeLRCommitment, the mkChallenges
accumulator of each round) and the final Bool are compared bit for bit."""
import os
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from util import golden, up, load_proof

pytestmark = pytest.mark.gpu
R = o.R


@pytest.fixture(scope="module")
def eng_gens(engine):
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    return engine


def pack_batch(proofs, vcommits, m2):
    """SoA byte arrays for bp_vbatch_load."""
    B = len(proofs)
    k = len(proofs[0]["productProof"]["lCommits"])
    sc = b"".join(b"".join(_lib.enc_scalar(f(pr)) for pr in proofs) for f in (
        lambda p: p["tBlinding"], lambda p: p["mu"], lambda p: p["t"],
        lambda p: p["productProof"]["l"], lambda p: p["productProof"]["r"]))
    rows = [[pr["aCommit"] for pr in proofs], [pr["sCommit"] for pr in proofs],
            [pr["t1Commit"] for pr in proofs], [pr["t2Commit"] for pr in proofs]]
    rows += [[pr["productProof"]["lCommits"][r] for pr in proofs] for r in range(k)]
    rows += [[pr["productProof"]["rCommits"][r] for pr in proofs] for r in range(k)]
    for j in range(m2):
        rows.append([(vc[j] if j < len(vc) else None) for vc in vcommits])
    pts = b"".join(_lib.enc_points(row) for row in rows)
    assert len(sc) == 5 * B * 32
    return sc, pts, k


def run_batch(eng, n, m2, proofs, vcommits, check_trace=True):
    B = len(proofs)
    sc, pts, k = pack_batch(proofs, vcommits, m2)
    vb = eng.vbatch(n, m2, B)
    assert vb.rounds == k
    vb.load(sc, pts)
    ys = [o.shamir_y(p["aCommit"], p["sCommit"]) for p in proofs]
    zs = [o.shamir_z(p["aCommit"], p["sCommit"], y) for p, y in zip(proofs, ys)]
    xs = [o.shamir_x(p["aCommit"], p["sCommit"], p["t1Commit"], p["t2Commit"], y, z)
          for p, y, z in zip(proofs, ys, zs)]
    us = [o.shamir_u(p["tBlinding"], p["mu"], p["t"]) for p in proofs]
    chal = _lib.enc_scalars(ys) + _lib.enc_scalars(zs) + _lib.enc_scalars(xs) + _lib.enc_scalars(us)
    acc = _lib.dec_points(vb.begin(chal), B)
    # check_trace: True (every proof), False, or the indices to compare with the oracle
    traced = range(B) if check_trace is True else (check_trace or ())
    if traced:
        for i in traced:
            p = proofs[i]
            hsp = o.hs_prime(ys[i], n * m2)
            exp = o.compute_lr_commitment(n, m2, p["aCommit"], p["sCommit"], p["t"], p["tBlinding"], p["mu"],
                                          xs[i], ys[i], zs[i], hsp)
            assert acc[i] == exp, "computeLRCommitment mismatch for proof %d" % i
    for r in range(k):
        xk = [o.shamir_x_prime(acc[i], p["productProof"]["lCommits"][r], p["productProof"]["rCommits"][r])
              for i, p in enumerate(proofs)]
        new = _lib.dec_points(vb.round(r, _lib.enc_scalars(xk)), B)
        if traced:
            for i in traced:
                p = proofs[i]
                li, ri = p["productProof"]["lCommits"][r], p["productProof"]["rCommits"][r]
                xi = pow(xk[i], -1, R)
                exp = o.point_add(o.point_add(o.point_mul(li, xk[i] * xk[i] % R), o.point_mul(ri, xi * xi % R)), acc[i])
                assert new[i] == exp, "mkChallenges accumulator mismatch, proof %d round %d" % (i, r)
        acc = new
    verdicts = vb.finish()
    vb.close()
    return verdicts


def mutate(rng, pr, vcs):
    """Corruptions in the spirit of test/Test/Protocol/RangeProof.hs:183-194 and SURVEY 8d config 4."""
    pr = dict(pr, productProof=dict(pr["productProof"]))
    kind = rng.randrange(6)
    if kind == 0:
        pr["t"] = (pr["t"] + 1) % R
    elif kind == 1 and pr["productProof"]["lCommits"]:
        ipp = pr["productProof"]
        ipp["lCommits"], ipp["rCommits"] = list(ipp["rCommits"]), list(ipp["lCommits"])
    elif kind == 2:
        vcs = [o.point_add(vcs[0], o.GEN)] + list(vcs[1:])  # commit (v+1) r
    elif kind == 3:
        pr["productProof"]["l"] = (pr["productProof"]["l"] + 1) % R
    elif kind == 4:
        pr["mu"] = (pr["mu"] + 5) % R
    else:
        pr["tBlinding"] = (pr["tBlinding"] + 1) % R
    return pr, vcs


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (8, 1), (4, 2), (8, 3), (16, 4)])
def test_state_machine_vs_oracle(eng_gens, n, m):
    rng = random.Random(1000 + 10 * n + m)
    m2 = 2 ** o.log2_ceil(m)
    proofs, vcs = [], []
    for _ in range(3):
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        pr = o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m2))
        proofs.append(pr)
        vcs.append([o.commit(v, b) for v, b in vs])
    # honest + mutated copies in one batch: verdicts are per proof
    bad = [mutate(rng, proofs[i % 3], vcs[i % 3]) for i in range(5)]
    all_p = proofs + [b[0] for b in bad]
    all_v = vcs + [b[1] for b in bad]
    got = run_batch(eng_gens, n, m2, all_p, all_v)
    exp = [o.mrp_verify_proof(2**n, v, p) for p, v in zip(all_p, all_v)]
    assert exp[:3] == [True] * 3
    assert got == exp


@pytest.mark.parametrize("n,m,copies", [(1, 1, 13), (4, 2, 13), (2, 4, 13), (2, 3, 13), (2, 8, 13), (1, 1, 1100),
                                        (2, 2, 1100)])
def test_throughput_kernels_with_aggregated_proofs(eng_gens, n, m, copies):
    """Above 96 proofs the state machine leaves the one-lane-per-bit path: x * S and the t-check's V_j, T1, T2 run
    on the joint tables of k_round_tables (verify.hip), the t-check with one or two tables per chain (m + 2 terms in
    pairs: m = 1 a pair and a single point, m = 4 three pairs in two passes); above 8,192 the stage-1 tail is the
    one-lane-per-proof kernel.  n * m = 1 has no rounds at all (tables of S and the t-check only).  Every stage's
    accumulator and the verdicts against the oracle, honest and corrupted proofs mixed."""
    rng = random.Random(3000 + 10 * n + m)
    m2 = 2 ** o.log2_ceil(m)
    proofs, vcs = [], []
    for _ in range(3):
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        proofs.append(o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m2)))
        vcs.append([o.commit(v, b) for v, b in vs])
    bad = [mutate(rng, proofs[i % 3], vcs[i % 3]) for i in range(5)]
    all_p = proofs + [b[0] for b in bad]
    all_v = vcs + [b[1] for b in bad]
    nb = len(all_p)
    exp = [o.mrp_verify_proof(2**n, v, p) for p, v in zip(all_p, all_v)]
    assert exp[:3] == [True] * 3
    got = run_batch(eng_gens, n, m2, all_p * copies, all_v * copies,
                    check_trace=list(range(nb)) + list(range(nb * (copies - 1), nb * copies)))
    assert got == exp * copies


def degenerate_lr(pr, kind, r):
    """A copy of the proof whose round-r commitments hit the group law's special cases."""
    ipp = dict(pr["productProof"])
    L, Rr = list(ipp["lCommits"]), list(ipp["rCommits"])
    if kind == 0:
        L[r] = None                       # L = O: identity entries in the round's joint table
    elif kind == 1:
        Rr[r] = None
    elif kind == 2:
        Rr[r] = L[r]                      # L + R is a doubling; the chain meets acc == entry
    elif kind == 3:
        Rr[r] = o.point_neg(L[r])         # L + R = O
    else:
        L[r] = Rr[r] = None
    ipp["lCommits"], ipp["rCommits"] = L, Rr
    return dict(pr, productProof=ipp)


@pytest.mark.parametrize("copies", [1, 50, 500])
def test_round_kernels_special_cases(eng_gens, copies):
    """Three forms of the rounds: one lane per scalar bit up to 768 proofs (17 here), two lanes per proof up to 8,192
    (850), one lane per proof above (8,500).  Every round's accumulator against the oracle for honest proofs and for
    proofs whose L_r / R_r are O, equal or opposite (the exceptional cases of the additions inside the chains)."""
    n, rng = 8, random.Random(77)
    base = []
    for _ in range(2):
        v, b = rng.randrange(0, 2**n), rng.randrange(1, R)
        base.append((o.mrp_generate_proof(2**n, [(v, b)], *o.draw_blinders(rng, n)), [o.commit(v, b)]))
    proofs = [p for p, _ in base]
    vcs = [v for _, v in base]
    for kind in range(5):
        for r in range(3):
            src = base[(kind + r) % 2]
            proofs.append(degenerate_lr(src[0], kind, r))
            vcs.append(src[1])
    nb = len(proofs)
    assert nb > 16
    exp = [o.mrp_verify_proof(2**n, v, p) for p, v in zip(proofs, vcs)]
    assert exp[:2] == [True, True]
    got = run_batch(eng_gens, n, 1, proofs * copies, vcs * copies,
                    check_trace=list(range(nb)) + list(range(nb * (copies - 1), nb * copies)))
    assert got == exp * copies


def test_bench_proof_golden(eng_gens):
    """The reference's bench input (bench/Bench/RangeProof.hs:10-14), n = 64."""
    b = golden()["bench_proof"]
    pr = load_proof(b["proof"])
    vc = up(b["vCommit"])
    rng = random.Random(5)
    batch_p = [pr] * 3 + [mutate(rng, pr, [vc])[0] for _ in range(3)]
    batch_v = [[vc]] * 6
    got = run_batch(eng_gens, 64, 1, batch_p, batch_v, check_trace=False)
    assert got[:3] == [True] * 3
    assert got[3:] == [o.rp_verify_proof(2**64, vc, p) for p in batch_p[3:]]
    # commitment to v+1 must not verify (RangeProof.hs:183-194)
    assert run_batch(eng_gens, 64, 1, [pr], [[o.commit(b["v"] + 1, b["vBlinding"])]], check_trace=False) == [False]


def test_derive_generators_matches_oracle(engine):
    """Utils.hs:14-25: h, gs, hs as the engine's host layer evaluates them."""
    g, hh, h1 = engine.derive_generators(8)
    assert h1 == o.h() and g == o.gs(8) and hh == o.hs(8)
    engine.set_generators(o.gs(128), o.hs(128), o.h())  # restore for the other tests


def test_small_batch_long_vectors(engine):
    """A few proofs with nm = 256 take the small-batch paths of the verifier (y^-i by square-and-multiply and
    the closed-form power sum, one lane per term in the fixed-base sums, rounds through eight lanes per proof):
    computeLRCommitment, every mkChallenges accumulator and the verdicts equal the oracle's."""
    rng = random.Random(4242)
    n, m = 64, 4
    engine.derive_generators(n * m)
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
    pr = engine.prove(n, vs, *o.draw_blinders(rng, n * m))
    vc = [o.commit(v, b) for v, b in vs]
    bad = dict(pr, t=(pr["t"] + 1) % R)
    assert run_batch(engine, n, m, [pr, bad], [vc, vc]) == [True, False]
    engine.set_generators(o.gs(128), o.hs(128), o.h())  # restore for the other tests


def test_generator_table_file(engine, tmp_path):
    """bp_generators_save / bp_generators_load: the file holds exactly the oracle's h, gs, hs (Utils.hs:14-25) with a
    checksum; a loaded table verifies proofs like a derived one; corrupted and off-curve files are refused."""
    import hashlib
    import struct
    n = 16
    engine.derive_generators(n)
    path = str(tmp_path / "gens.bin")
    engine.save_generators(path)
    raw = open(path, "rb").read()
    body = raw[:-32]
    assert body[:8] == b"BPGENS01" and struct.unpack("<4I", body[8:24]) == (n, 0, 0, 0)
    assert body[24:] == _lib.enc_points(o.gs(n)) + _lib.enc_points(o.hs(n)) + _lib.enc_point(o.h())
    assert raw[-32:] == hashlib.sha256(body).digest()
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    engine.load_generators(path)
    rng = random.Random(5)
    vs = [(rng.randrange(0, 2**8), rng.randrange(1, R)) for _ in range(2)]
    pr = o.mrp_generate_proof(2**8, vs, *o.draw_blinders(rng, 16))
    vc = [o.commit(v, b) for v, b in vs]
    sc, pts, _ = pack_batch([pr, dict(pr, mu=(pr["mu"] + 1) % R)], [vc, vc], 2)
    assert engine.verify_batch(8, 2, 2, sc, pts) == [True, False]
    # a flipped byte fails the checksum; an off-curve entry with a matching checksum fails the curve check
    bad = bytearray(raw)
    bad[200] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(_lib.BpError):
        engine.load_generators(path)
    off = bytearray(body)
    off[24 + 32] ^= 1  # y of gs_1
    open(path, "wb").write(bytes(off) + hashlib.sha256(bytes(off)).digest())
    with pytest.raises(_lib.BpError):
        engine.load_generators(path)
    with pytest.raises(_lib.BpError):
        engine.load_generators(str(tmp_path / "missing.bin"))
    engine.set_generators(o.gs(128), o.hs(128), o.h())  # restore for the other tests


def test_host_layer_verify_batch(eng_gens):
    """bp_rangeproof_verify_batch = map verifyProof, transcript hashed by the C++ host layer;
    chunked ping-pong pipeline must give the same verdicts as one chunk."""
    rng = random.Random(77)
    n, m = 8, 2
    proofs, vcs = [], []
    for _ in range(4):
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        proofs.append(o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m)))
        vcs.append([o.commit(v, b) for v, b in vs])
    all_p, all_v = [], []
    for i in range(37):
        if i % 3 == 0:
            p, v = mutate(rng, proofs[i % 4], vcs[i % 4])
        else:
            p, v = proofs[i % 4], vcs[i % 4]
        all_p.append(p)
        all_v.append(v)
    exp = [o.mrp_verify_proof(2**n, v, p) for p, v in zip(all_p, all_v)]
    sc, pts, _ = pack_batch(all_p, all_v, m)
    assert eng_gens.verify_batch(n, m, len(all_p), sc, pts) == exp
    for chunks, threads in [(1, 1), (3, 2), (5, 4)]:
        job = eng_gens.vjob(n, m, len(all_p), chunks, threads)
        job.load(sc, pts)
        assert [b != 0 for b in job.run()] == exp
        assert [b != 0 for b in job.run()] == exp  # re-runnable
        job.close()
    # chunks of unequal size (BP_CHUNK_WEIGHTS, read when the job is created): sizes are rounded to 256 proofs and
    # clamped so that every chunk keeps a proof - here 34, 1, 1, 1 and 35, 1, 1
    for weights, chunks in [("1,5,0,2", 4), ("3,1,1", 3)]:
        os.environ["BP_CHUNK_WEIGHTS"] = weights
        try:
            job = eng_gens.vjob(n, m, len(all_p), chunks, 2)
        finally:
            del os.environ["BP_CHUNK_WEIGHTS"]
        job.load(sc, pts)
        assert [b != 0 for b in job.run()] == exp
        job.close()


def test_invalid_encodings_do_not_verify(eng_gens):
    """Off-curve points make that proof (and only that proof) fail; O stays a legal point (padding commitments,
    Verifier.hs:37-38, exercised by the m = 3 cases above).  Coordinates >= p and scalars >= r are refused by the same
    kernel; honest 256-bit values leave no room to build such an encoding from a valid proof, so that arm is
    covered by the generator test below (canonical check) only."""
    b = golden()["bench_proof"]
    pr, vc = load_proof(b["proof"]), up(b["vCommit"])
    x, y = pr["t1Commit"]
    off_curve = dict(pr, t1Commit=(x, (y + 1) % o.P))
    ipp = dict(pr["productProof"])
    ipp["lCommits"] = [(ipp["lCommits"][0][0], (ipp["lCommits"][0][1] + 5) % o.P)] + list(ipp["lCommits"][1:])
    bad_l = dict(pr, productProof=ipp)
    proofs = [pr, off_curve, bad_l, pr]
    sc, pts, _ = pack_batch(proofs, [[vc]] * len(proofs), 1)
    assert eng_gens.verify_batch(64, 1, len(proofs), sc, pts) == [True, False, False, True]
    # an off-curve V fails only through the check (verifyTPoly would otherwise run on garbage)
    sc, pts, _ = pack_batch([pr, pr], [[(vc[0], (vc[1] + 1) % o.P)], [vc]], 1)
    assert eng_gens.verify_batch(64, 1, 2, sc, pts) == [False, True]


def test_generators_are_validated_and_replacing_them_invalidates_batches(engine):
    """bp_ctx_set_generators refuses points that are not finite curve points (and leaves the context without
    generators, not with half a table); a batch object created before the generators were replaced is refused."""
    gs, hs, h = o.gs(4), o.hs(4), o.h()
    engine.set_generators(gs, hs, h)
    vb = engine.vbatch(4, 1, 1)
    bad = list(gs)
    bad[2] = (bad[2][0], (bad[2][1] + 1) % o.P)
    with pytest.raises(_lib.BpError):
        engine.set_generators(bad, hs, h)
    with pytest.raises(_lib.BpError):
        engine.set_generators(gs, hs, None)          # h = O
    with pytest.raises(_lib.BpError):
        engine.vbatch(4, 1, 1)                       # no generators installed after the failed calls
    engine.set_generators(gs, hs, h)
    rng = random.Random(3)
    v, g = 9, rng.randrange(1, R)
    pr = o.mrp_generate_proof(2**4, [(v, g)], *o.draw_blinders(rng, 4))
    sc, pts, _ = pack_batch([pr], [[o.commit(v, g)]], 1)
    with pytest.raises(_lib.BpError):
        vb.load(sc, pts)                             # created against the old generator set
    vb.close()
    assert engine.verify_batch(4, 1, 1, sc, pts) == [True]
    engine.set_generators(o.gs(128), o.hs(128), o.h())  # restore for the other tests


def test_verify_batch_job_cache(eng_gens):
    """bp_rangeproof_verify_batch keeps its job on the context: same shape again (other proofs), another shape, the
    first shape again - always the oracle's verdicts."""
    rng = random.Random(91)

    def make(n, count):
        ps, vs = [], []
        for i in range(count):
            v, g = rng.randrange(2**n), rng.randrange(1, R)
            p = o.mrp_generate_proof(2**n, [(v, g)], *o.draw_blinders(rng, n))
            c = o.commit(v, g)
            if i % 2:
                p = dict(p, t=(p["t"] + 1) % R)
            ps.append(p)
            vs.append([c])
        return ps, vs
    for n, count in [(8, 3), (8, 3), (4, 2), (8, 3), (8, 5)]:
        ps, vs = make(n, count)
        sc, pts, _ = pack_batch(ps, vs, 1)
        assert eng_gens.verify_batch(n, 1, count, sc, pts) == [i % 2 == 0 for i in range(count)]
