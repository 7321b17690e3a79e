"""Generates tests/golden/config2_vectors.json from the oracle (oracle/bp_oracle.py): BASELINE.json configs[1],
MultiRangeProof aggregated m = 16 x 64-bit (n*m = 1024, 2048 generators, 10 inner-product rounds), seed 2 with
explicit blinders.  Like vectors.json these pin the ORACLE's outputs: the reference holds no
known-answer vectors of its own.  Recorded: the inputs, the proof (MultiRangeProof/Prover.hs:47-123), the verifier's
intermediates (computeLRCommitment, every mkChallenges accumulator; MultiRangeProof/Verifier.hs:19-88,
InnerProductProof/Verifier.hs:40-53) and the verdicts of the honest proof and of three mutations.
Takes about two minutes of pure-Python big-int work.  Re-run:  python tests/golden/make_golden_config2.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bp_oracle as o  # noqa: E402
from make_golden import hx, hp, proof_json  # noqa: E402

N, M = 64, 16


def mutations(pr, vcs):
    """The three corruption kinds of bench.py (flip t, swap L_0/R_0, wrong V)."""
    ipp = pr["productProof"]
    a = dict(pr, t=(pr["t"] + 1) % o.R)
    sw = dict(ipp, lCommits=[ipp["rCommits"][0]] + list(ipp["lCommits"][1:]),
              rCommits=[ipp["lCommits"][0]] + list(ipp["rCommits"][1:]))
    b = dict(pr, productProof=sw)
    c_v = [vcs[1]] + list(vcs[1:])
    return [("t+1", a, vcs), ("swap L0 R0", b, vcs), ("V0 := V1", pr, c_v)]


def main():
    rng = random.Random(2)
    vs = [(rng.randrange(0, 2**N), rng.randrange(1, o.R)) for _ in range(M)]
    bl = o.draw_blinders(rng, N * M)
    trace = {}
    pr = o.mrp_generate_proof_unsafe(2**N, vs, *bl, trace=trace)
    vcs = [o.commit(v, b) for v, b in vs]
    ipp = pr["productProof"]
    # verifier side, recomputed independently of the prover's trace
    y = o.shamir_y(pr["aCommit"], pr["sCommit"])
    z = o.shamir_z(pr["aCommit"], pr["sCommit"], y)
    x = o.shamir_x(pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"], y, z)
    u = o.shamir_u(pr["tBlinding"], pr["mu"], pr["t"])
    assert (y, z, x) == (trace["y"], trace["z"], trace["x"])
    hsp = o.hs_prime(y, N * M)
    P = o.compute_lr_commitment(N, M, pr["aCommit"], pr["sCommit"], pr["t"], pr["tBlinding"], pr["mu"], x, y, z, hsp)
    assert P == trace["P"]
    accs, xks, acc = [], [], P
    for li, ri in zip(ipp["lCommits"], ipp["rCommits"]):
        xk = o.shamir_x_prime(acc, li, ri)
        xi = pow(xk, -1, o.R)
        acc = o.point_add(o.point_add(o.point_mul(li, xk * xk % o.R), o.point_mul(ri, xi * xi % o.R)), acc)
        xks.append(xk)
        accs.append(acc)
    assert xks == [r["x"] for r in trace["rounds"]]
    assert o.mrp_verify_proof(2**N, vcs, pr)
    muts = []
    for name, p2, v2 in mutations(pr, vcs):
        muts.append(dict(kind=name, verifies=o.mrp_verify_proof(2**N, v2, p2)))
        print("mutation", name, muts[-1]["verifies"], flush=True)
    out = dict(upperBound="2^64", n=N, m=M, seed=2,
               vs=[[hx(v), hx(b)] for v, b in vs],
               blinders=dict(sL=[hx(v) for v in bl[0]], sR=[hx(v) for v in bl[1]], aBlinding=hx(bl[2]),
                             sBlinding=hx(bl[3]), t1Blinding=hx(bl[4]), t2Blinding=hx(bl[5])),
               vCommits=[hp(p) for p in vcs], proof=proof_json(pr),
               challenges=dict(y=hx(y), z=hx(z), x=hx(x), u=hx(u), rounds=[hx(v) for v in xks]),
               P=hp(P), accumulators=[hp(a) for a in accs], verifies=True, mutations=muts)
    with open(os.path.join(HERE, "config2_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote config2_vectors.json")


if __name__ == "__main__":
    main()
