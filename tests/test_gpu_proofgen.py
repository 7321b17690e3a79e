"""GPU tests of the bench input generator (benchmarks/proofgen.hip): its proofs must be
byte-identical to the oracle's literal prover (MultiRangeProof/Prover.hs:47-123) for the same
(v, gamma) and blinders, and a full-size batch must verify with per-proof verdicts."""
import os
import sys

import numpy as np
import pytest

import bp_oracle as o
from bulletproofs_amd import _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))

pytestmark = pytest.mark.gpu


def le(a):
    return int.from_bytes(bytes(a), "little")


@pytest.mark.parametrize("n,B", [(8, 5), (64, 2)])
def test_generator_equals_literal_prover(engine, n, B):
    import proofgen
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    inp = proofgen.random_inputs(99 + n, n, B)
    sc, pts = proofgen.make_proofs(engine, _lib.enc_point(o.h()), n, B, inp, threads=2)
    k = n.bit_length() - 1
    for p in range(B):
        v, gamma = le(inp["v"][p]), le(inp["gamma"][p])
        sL = [le(inp["sL"][p, i]) for i in range(n)]
        sR = [le(inp["sR"][p, i]) for i in range(n)]
        pr = o.mrp_generate_proof_unsafe(2**n, [(v, gamma)], sL, sR, le(inp["alpha"][p]), le(inp["rho"][p]),
                                         le(inp["tau1"][p]), le(inp["tau2"][p]))
        ipp = pr["productProof"]
        exp_sc = [pr["tBlinding"], pr["mu"], pr["t"], ipp["l"], ipp["r"]]
        assert [le(sc[r, p]) for r in range(5)] == exp_sc
        exp_pts = [pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"]] + ipp["lCommits"] + \
            ipp["rCommits"] + [o.commit(v, gamma)]
        assert [_lib.dec_point(bytes(pts[r, p])) for r in range(4 + 2 * k + 1)] == exp_pts


def test_batch_of_generated_proofs_verifies(engine):
    import proofgen
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    n, B = 64, 3000
    inp = proofgen.random_inputs(7, n, B)
    sc, pts = proofgen.make_proofs(engine, _lib.enc_point(o.h()), n, B, inp, threads=4)
    rng = np.random.default_rng(1)
    bad = rng.choice(B, size=40, replace=False)
    sc[2, bad, 0] ^= 1
    exp = np.ones(B, dtype=bool)
    exp[bad] = False
    got = np.array(engine.verify_batch(n, 1, B, sc.tobytes(), pts.tobytes()))
    assert (got == exp).all()


@pytest.mark.parametrize("transcript", ["host", "device", "mixed"])
def test_two_jobs_in_flight_give_the_same_verdicts(engine, transcript, monkeypatch):
    """bp_vjob_submit / bp_vjob_wait: two jobs overlapping on one context return exactly what bp_vjob_run returns for
    each of them, pass after pass (different batches, different corrupted proofs, chunked and single-chunk jobs), and
    the state rules hold (no second pass, load or run while one is in flight; no wait without submit).  With the
    transcript hashed by the host, formed on the device (BP_FS_DEVICE=1 when the job is created), and one job of each
    form side by side."""
    import ctypes
    import proofgen
    forms = {"host": ("0", "0"), "device": ("1", "1"), "mixed": ("1", "0")}[transcript]
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    n = 64
    jobs, expect = [], []
    for seed, B, chunks in ((11, 9000, 2), (12, 5000, 1)):
        inp = proofgen.random_inputs(seed, n, B)
        sc, pts = proofgen.make_proofs(engine, _lib.enc_point(o.h()), n, B, inp, threads=4)
        bad = np.random.default_rng(seed).choice(B, size=B // 50, replace=False)
        sc[2, bad, 0] ^= 1
        exp = np.ones(B, dtype=np.uint8)
        exp[bad] = 0
        monkeypatch.setenv("BP_FS_DEVICE", forms[len(jobs)])
        job = engine.vjob(n, 1, B, chunks, 4)
        job.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))
        assert (np.frombuffer(job.run(), dtype=np.uint8) == exp).all()
        jobs.append((job, sc, pts))
        expect.append(exp)
    a, b = jobs[0][0], jobs[1][0]
    with pytest.raises(_lib.BpError):
        a.wait()                                # nothing submitted
    a.submit()
    b.submit()
    with pytest.raises(_lib.BpError):
        a.submit()                              # one pass per job at a time
    with pytest.raises(_lib.BpError):
        a.run()
    with pytest.raises(_lib.BpError):
        a.load(jobs[0][1].ctypes.data_as(ctypes.c_char_p), jobs[0][2].ctypes.data_as(ctypes.c_char_p))
    for _ in range(3):                          # steady state: a new pass of A is submitted while B is still running
        va = np.frombuffer(a.wait(), dtype=np.uint8)
        a.submit()
        vb = np.frombuffer(b.wait(), dtype=np.uint8)
        b.submit()
        assert (va == expect[0]).all() and (vb == expect[1]).all()
    assert (np.frombuffer(a.wait(), dtype=np.uint8) == expect[0]).all()
    assert (np.frombuffer(b.wait(), dtype=np.uint8) == expect[1]).all()
    assert (np.frombuffer(a.run(), dtype=np.uint8) == expect[0]).all()   # the synchronous form still works afterwards
    b.submit()
    b.close()                                   # destroying a job joins its pass first
    a.close()
