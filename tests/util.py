"""Shared helpers for the tests."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)


def ux(h):
    return int(h, 16)


def up(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def load_proof(pj):
    return dict(tBlinding=ux(pj["tBlinding"]), mu=ux(pj["mu"]), t=ux(pj["t"]), aCommit=up(pj["aCommit"]),
                sCommit=up(pj["sCommit"]), t1Commit=up(pj["t1Commit"]), t2Commit=up(pj["t2Commit"]),
                productProof=dict(lCommits=[up(p) for p in pj["lCommits"]], rCommits=[up(p) for p in pj["rCommits"]],
                                  l=ux(pj["l"]), r=ux(pj["r"])))


def load_blinders(bj):
    return ([ux(v) for v in bj["sL"]], [ux(v) for v in bj["sR"]], ux(bj["aBlinding"]), ux(bj["sBlinding"]),
            ux(bj["t1Blinding"]), ux(bj["t2Blinding"]))


def circuit_golden():
    with open(os.path.join(HERE, "golden", "circuit_vectors.json")) as f:
        return json.load(f)


def load_circuit_case(c):
    """-> (circuit, witness, blinders tuple, proof) in the oracle's dict shapes."""
    rows = lambda m: [[ux(v) for v in row] for row in m]
    cj = c["circuit"]
    circuit = dict(wL=rows(cj["wL"]), wR=rows(cj["wR"]), wO=rows(cj["wO"]),
                   commitmentWeights=rows(cj["commitmentWeights"]), cs=[ux(v) for v in cj["cs"]])
    witness = dict(assignment={k: [ux(v) for v in c["assignment"][k]] for k in ("aL", "aR", "aO")},
                   commitments=[up(p) for p in c["commitments"]], commitBlinders=[ux(v) for v in c["commitBlinders"]])
    b = c["blinders"]
    blinders = (ux(b["aiBlinding"]), ux(b["aoBlinding"]), ux(b["sBlinding"]), [ux(v) for v in b["sL"]],
                [ux(v) for v in b["sR"]], [ux(v) for v in b["tBlindings"]])
    pj = c["proof"]
    proof = dict(tBlinding=ux(pj["tBlinding"]), mu=ux(pj["mu"]), t=ux(pj["t"]), aiCommit=up(pj["aiCommit"]),
                 aoCommit=up(pj["aoCommit"]), sCommit=up(pj["sCommit"]), tCommits=[up(p) for p in pj["tCommits"]],
                 productProof=dict(lCommits=[up(p) for p in pj["lCommits"]], rCommits=[up(p) for p in pj["rCommits"]],
                                   l=ux(pj["l"]), r=ux(pj["r"])))
    return circuit, witness, blinders, proof


def config5_inputs(o, n=4096, seed=5):
    """BASELINE config 5: the Hadamard-style circuit of test/Test/Protocol/ArithCircuit.hs:48-83 generalised
    to n = 2^12 multiplication gates with Q = 2 dense rows (sum aL = v0, sum aR = v1), aL, aR uniform in Fr, seed 5.
    -> (circuit, witness without the commitments, commitment openings [(v0, r), (v1, s)], blinders) in the oracle's
    shapes.  Shared by tests/golden/make_golden_config5.py and the GPU test so that the fixture holds outputs only."""
    import random
    R = o.R
    rng = random.Random(seed)
    a_l = [rng.randrange(R) for _ in range(n)]
    a_r = [rng.randrange(R) for _ in range(n)]
    r, s = rng.randrange(R), rng.randrange(R)
    circuit = dict(wL=[[1] * n, [0] * n], wR=[[0] * n, [1] * n], wO=[[0] * n, [0] * n],
                   commitmentWeights=[[1, 0], [0, 1]], cs=[0, 0])
    openings = [(sum(a_l) % R, r), (sum(a_r) % R, s)]
    witness = dict(assignment=dict(aL=a_l, aR=a_r, aO=o.hadamard(a_l, a_r)), commitBlinders=[r, s])
    three = [rng.randrange(1, R) for _ in range(3)]
    s_l = [rng.randrange(1, R) for _ in range(n)]
    s_r = [rng.randrange(1, R) for _ in range(n)]
    five = [rng.randrange(1, R) for _ in range(5)]
    return circuit, witness, openings, (*three, s_l, s_r, five)


def config5_mutations(o, proof, commitments):
    """Three mutations of the config-5 proof: t + 1, swapped commitments, L_0 <-> R_0."""
    R = o.R
    ipp = proof["productProof"]
    swapped = dict(ipp, lCommits=[ipp["rCommits"][0]] + list(ipp["lCommits"][1:]),
                   rCommits=[ipp["lCommits"][0]] + list(ipp["rCommits"][1:]))
    return [(dict(proof, t=(proof["t"] + 1) % R), commitments),
            (proof, [commitments[1], commitments[0]]),
            (dict(proof, productProof=swapped), commitments)]
