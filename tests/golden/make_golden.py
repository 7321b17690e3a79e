"""Generates the committed golden vectors under tests/golden/ from the oracle
(oracle/bp_oracle.py).  The reference holds no known-answer vectors of its own
(all of its tests are QuickCheck properties), so these pin the ORACLE's outputs
on fixed seeded inputs; the oracle itself is pinned against OpenSSL's secp256k1
in tests/test_oracle.py.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bp_oracle as o  # noqa: E402


def hx(v):
    return "%064x" % v


def hp(pt):
    return None if pt is None else [hx(pt[0]), hx(pt[1])]


def proof_json(pr):
    ipp = pr["productProof"]
    return dict(tBlinding=hx(pr["tBlinding"]), mu=hx(pr["mu"]), t=hx(pr["t"]), aCommit=hp(pr["aCommit"]),
                sCommit=hp(pr["sCommit"]), t1Commit=hp(pr["t1Commit"]), t2Commit=hp(pr["t2Commit"]),
                lCommits=[hp(p) for p in ipp["lCommits"]], rCommits=[hp(p) for p in ipp["rCommits"]],
                l=hx(ipp["l"]), r=hx(ipp["r"]))


def main():
    rng = random.Random(20240611)
    out = {}
    # generators (U1 = derived Show, U2 = p34 root)
    out["h"] = hp(o.h())
    out["gs"] = [hp(p) for p in o.gs(4)]
    out["hs"] = [hp(p) for p in o.hs(4)]
    # field vectors
    edge_q = [0, 1, 2, o.P - 1, o.P - 2, 2**255, 2**32 + 977, 977]
    edge_r = [0, 1, 2, o.R - 1, o.R - 2, 2**255, 2**128]
    fq, fr = [], []
    for _ in range(24):
        a = rng.choice(edge_q + [rng.randrange(o.P)] * 3)
        b = rng.choice(edge_q + [rng.randrange(o.P)] * 3)
        fq.append(dict(a=hx(a), b=hx(b), add=hx((a + b) % o.P), sub=hx((a - b) % o.P), mul=hx(a * b % o.P),
                       inv=hx(pow(a, -1, o.P) if a else 0)))
        a = rng.choice(edge_r + [rng.randrange(o.R)] * 3)
        b = rng.choice(edge_r + [rng.randrange(o.R)] * 3)
        fr.append(dict(a=hx(a), b=hx(b), add=hx((a + b) % o.R), sub=hx((a - b) % o.R), mul=hx(a * b % o.R),
                       inv=hx(pow(a, -1, o.R) if a else 0)))
    out["fq"] = fq
    out["fr"] = fr
    # k*G and point sums
    ks = [1, 2, 3, o.R - 1, 2**128, rng.randrange(o.R), rng.randrange(o.R)]
    out["kG"] = [dict(k=hx(k), P=hp(o.point_mul(o.GEN, k))) for k in ks]
    # sumExps cases (incl. empty, identity points, zero scalars, repeated points)
    pts = [o.point_mul(o.GEN, rng.randrange(1, o.R)) for _ in range(12)]
    cases = []
    for n in [0, 1, 2, 3, 7, 12, 40]:
        ps = [rng.choice(pts + [None]) for _ in range(n)]
        es = [rng.choice([0, 1, o.R - 1, rng.randrange(o.R), rng.randrange(o.R), rng.randrange(2**64)])
              for _ in range(n)]
        cases.append(dict(scalars=[hx(e) for e in es], points=[hp(p) for p in ps], result=hp(o.sum_exps(es, ps))))
    out["sumExps"] = cases
    # the reference's own bench input: upperBound = 2^64, (v, gamma) = (7238283, 827361)
    # (bench/Bench/RangeProof.hs:10-14); blinders from a seeded stand-in for MonadRandom (U3)
    brng = random.Random(1)
    bl = o.draw_blinders(brng, 64)
    trace = {}
    pr = o.mrp_generate_proof_unsafe(2**64, [(7238283, 827361)], *bl, trace=trace)
    vc = o.commit(7238283, 827361)
    assert o.rp_verify_proof(2**64, vc, pr)
    out["bench_proof"] = dict(upperBound="2^64", v=7238283, vBlinding=827361,
                              blinders=dict(sL=[hx(v) for v in bl[0]], sR=[hx(v) for v in bl[1]], aBlinding=hx(bl[2]),
                                            sBlinding=hx(bl[3]), t1Blinding=hx(bl[4]), t2Blinding=hx(bl[5])),
                              vCommit=hp(vc), proof=proof_json(pr),
                              challenges=dict(y=hx(trace["y"]), z=hx(trace["z"]), x=hx(trace["x"]),
                                              rounds=[hx(r["x"]) for r in trace["rounds"]]),
                              P=hp(trace["P"]), verifies=True)
    # a small aggregated proof: n = 8, m = 3 (pads to 4)
    brng = random.Random(2)
    vs = [(brng.randrange(256), brng.randrange(1, o.R)) for _ in range(3)]
    bl = o.draw_blinders(brng, 8 * 4)
    pr = o.mrp_generate_proof(2**8, vs, *bl)
    vcs = [o.commit(v, b) for v, b in vs]
    assert o.mrp_verify_proof(2**8, vcs, pr)
    out["multi_proof"] = dict(upperBound="2^8", vs=[[hx(v), hx(b)] for v, b in vs],
                              blinders=dict(sL=[hx(v) for v in bl[0]], sR=[hx(v) for v in bl[1]], aBlinding=hx(bl[2]),
                                            sBlinding=hx(bl[3]), t1Blinding=hx(bl[4]), t2Blinding=hx(bl[5])),
                              vCommits=[hp(p) for p in vcs], proof=proof_json(pr), verifies=True)
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote vectors.json")
    circuit_vectors()


def ac_proof_json(pr):
    ipp = pr["productProof"]
    return dict(tBlinding=hx(pr["tBlinding"]), mu=hx(pr["mu"]), t=hx(pr["t"]), aiCommit=hp(pr["aiCommit"]),
                aoCommit=hp(pr["aoCommit"]), sCommit=hp(pr["sCommit"]), tCommits=[hp(p) for p in pr["tCommits"]],
                lCommits=[hp(p) for p in ipp["lCommits"]], rCommits=[hp(p) for p in ipp["rCommits"]],
                l=hx(ipp["l"]), r=hx(ipp["r"]))


def circuit_vectors():
    """Arithmetic-circuit fixtures: the fixed circuits of the reference's own suite
    (test/Test/Protocol/ArithCircuit.hs:48-83 Hadamard with n = 4 here, :167-207 shuffle) with seeded
    witnesses and blinders; written to circuit_vectors.json (vectors.json stays untouched)."""
    R = o.R
    rng = random.Random(50)
    out = {}

    def record(name, circuit, witness, n2):
        bl = o.ac_draw_blinders(rng, n2)
        pr = o.ac_generate_proof(circuit, witness, *bl)
        assert o.ac_verify_proof(witness["commitments"], pr, circuit)
        fx = lambda rows: [[hx(v % R) for v in row] for row in rows]
        out[name] = dict(
            circuit=dict(wL=fx(circuit["wL"]), wR=fx(circuit["wR"]), wO=fx(circuit["wO"]),
                         commitmentWeights=fx(circuit["commitmentWeights"]), cs=[hx(v % R) for v in circuit["cs"]]),
            assignment={k: [hx(v % R) for v in witness["assignment"][k]] for k in ("aL", "aR", "aO")},
            commitBlinders=[hx(v) for v in witness["commitBlinders"]],
            commitments=[hp(p) for p in witness["commitments"]],
            blinders=dict(aiBlinding=hx(bl[0]), aoBlinding=hx(bl[1]), sBlinding=hx(bl[2]), sL=[hx(v) for v in bl[3]],
                          sR=[hx(v) for v in bl[4]], tBlindings=[hx(v) for v in bl[5]]),
            proof=ac_proof_json(pr), verifies=True)

    n = 4
    a_l = [rng.randrange(R) for _ in range(n)]
    a_r = [rng.randrange(R) for _ in range(n)]
    r, s2 = rng.randrange(R), rng.randrange(R)
    circuit = dict(wL=[[1] * n, [0] * n], wR=[[0] * n, [1] * n], wO=[[0] * n, [0] * n],
                   commitmentWeights=[[1, 0], [0, 1]], cs=[0, 0])
    witness = dict(assignment=dict(aL=a_l, aR=a_r, aO=o.hadamard(a_l, a_r)),
                   commitments=[o.commit(sum(a_l) % R, r), o.commit(sum(a_r) % R, s2)], commitBlinders=[r, s2])
    record("hadamard4", circuit, witness, 4)

    z = rng.randrange(R)
    circuit = dict(wL=[[0, 0], [1, 0], [0, 1], [0, 0], [0, 0]], wR=[[0, 0], [0, 0], [0, 0], [1, 0], [0, 1]],
                   wO=[[1, -1], [0, 0], [0, 0], [0, 0], [0, 0]],
                   commitmentWeights=[[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
                   cs=[0, -z, -z, -z, -z])
    a_l = [(4 - z) % R, (9 - z) % R]
    a_r = [(9 - z) % R, (4 - z) % R]
    bl = [rng.randrange(R) for _ in range(4)]
    witness = dict(assignment=dict(aL=a_l, aR=a_r, aO=o.hadamard(a_l, a_r)),
                   commitments=[o.commit(v, b) for v, b in zip([4, 9, 9, 4], bl)], commitBlinders=bl)
    record("shuffle", circuit, witness, 2)
    with open(os.path.join(HERE, "circuit_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote circuit_vectors.json")


if __name__ == "__main__":
    main()
