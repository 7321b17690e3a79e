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
