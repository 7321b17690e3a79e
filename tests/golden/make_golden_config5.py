"""Writes tests/golden/config5_vectors.json: BASELINE config 5 at FULL size (arithmetic circuit, n = 2^12
multiplication gates, Q = 2) evaluated by the CPU oracle (oracle/bp_oracle.py, the restatement of
ArithmeticCircuit/Prover.hs:16-117 and Verifier.hs:16-71).  The inputs are regenerated from the seed by
tests/util.py::config5_inputs, so the fixture holds outputs only: the challenges y, z, x, commitmentLR, u, the whole
proof (5 scalars + 34 points), a SHA-256 over the proof's canonical point bytes, and the oracle's verdicts for the
honest proof and three mutations.  Takes about ten minutes of pure-Python big-integer arithmetic:
    python tests/golden/make_golden_config5.py
The fixture is data; the oracle never travels into the product path."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import bp_oracle as o  # noqa: E402
from util import config5_inputs, config5_mutations  # noqa: E402


def hx(v):
    return "%064x" % v


def hp(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    t0 = time.time()
    n = 4096
    circuit, witness, openings, blinders = config5_inputs(o, n)
    commitments = [o.commit(v, b) for v, b in openings]
    witness = dict(witness, commitments=commitments)
    trace = {}
    proof = o.ac_generate_proof(circuit, witness, *blinders, trace=trace)
    print("proved in %.0f s" % (time.time() - t0), flush=True)
    ipp = proof["productProof"]
    pts = [proof["aiCommit"], proof["aoCommit"], proof["sCommit"]] + list(proof["tCommits"]) \
        + list(ipp["lCommits"]) + list(ipp["rCommits"])
    assert len(pts) == 34 and len(ipp["lCommits"]) == 12
    digest = hashlib.sha256(b"".join(o.wire_point(p) for p in pts)).hexdigest()
    verdicts = [o.ac_verify_proof(commitments, proof, circuit)]
    print("verified in %.0f s: %s" % (time.time() - t0, verdicts[0]), flush=True)
    for bad_proof, bad_commits in config5_mutations(o, proof, commitments):
        verdicts.append(o.ac_verify_proof(bad_commits, bad_proof, circuit))
        print("mutation -> %s (%.0f s)" % (verdicts[-1], time.time() - t0), flush=True)
    assert verdicts == [True, False, False, False]
    out = dict(
        note="config 5, n = 4096 gates, Q = 2, seed 5; inputs: tests/util.py::config5_inputs; generator: make_golden_config5.py",
        show_style=o.SHOW_STYLE if hasattr(o, "SHOW_STYLE") else 0,
        n=n, seed=5,
        commitments=[hp(p) for p in commitments],
        y=hx(trace["y"]), z=hx(trace["z"]), x=hx(trace["x"]), commitmentLR=hp(trace["P"]), u=hp(trace["u"]),
        proof=dict(tBlinding=hx(proof["tBlinding"]), mu=hx(proof["mu"]), t=hx(proof["t"]), aiCommit=hp(proof["aiCommit"]),
                   aoCommit=hp(proof["aoCommit"]), sCommit=hp(proof["sCommit"]), tCommits=[hp(p) for p in proof["tCommits"]],
                   lCommits=[hp(p) for p in ipp["lCommits"]], rCommits=[hp(p) for p in ipp["rCommits"]],
                   l=hx(ipp["l"]), r=hx(ipp["r"])),
        points_sha256=digest,
        verdicts=dict(honest=verdicts[0], t_plus_1=verdicts[1], commitments_swapped=verdicts[2], l0_r0_swapped=verdicts[3]),
    )
    with open(os.path.join(HERE, "config5_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote config5_vectors.json in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
