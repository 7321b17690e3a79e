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
