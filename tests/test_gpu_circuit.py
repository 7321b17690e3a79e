"""GPU parity tests for the arithmetic-circuit protocol: bp_ac_prove / bp_ac_verify
against the oracle's restatement of ArithmeticCircuit/{Prover,Verifier,Internal}.hs on the circuits
of the reference's own suite (test/Test/Protocol/ArithCircuit.hs), bit for bit with shared blinders."""
import random

import pytest

import bp_oracle as o

pytestmark = pytest.mark.gpu
R = o.R


@pytest.fixture(scope="module")
def eng_gens(engine):
    engine.set_generators(o.gs(128), o.hs(128), o.h())
    return engine


def hadamard_circuit(rng, n):
    """test_arithCircuitProof_hadamard (ArithCircuit.hs:48-83): sum aL = v0, sum aR = v1."""
    a_l = [rng.randrange(R) for _ in range(n)]
    a_r = [rng.randrange(R) for _ in range(n)]
    r, s = rng.randrange(R), rng.randrange(R)
    circuit = dict(wL=[[1] * n, [0] * n], wR=[[0] * n, [1] * n], wO=[[0] * n, [0] * n],
                   commitmentWeights=[[1, 0], [0, 1]], cs=[0, 0])
    witness = dict(assignment=dict(aL=a_l, aR=a_r, aO=o.hadamard(a_l, a_r)),
                   commitments=[o.commit(sum(a_l) % R, r), o.commit(sum(a_r) % R, s)], commitBlinders=[r, s])
    return circuit, witness


def no_mult_gates(rng):
    """ArithCircuit.hs:92-117: n = 0, m = 3, q = 1."""
    bl = [rng.randrange(R) for _ in range(3)]
    circuit = dict(wL=[[]], wR=[[]], wO=[[]], commitmentWeights=[[1, 1, -1]], cs=[0])
    witness = dict(assignment=dict(aL=[], aR=[], aO=[]),
                   commitments=[o.commit(v, b) for v, b in zip([2, 5, 7], bl)], commitBlinders=bl)
    return circuit, witness


def no_input_values(rng):
    """ArithCircuit.hs:130-153: n = 1, m = 0, q = 3."""
    circuit = dict(wL=[[0], [0], [1]], wR=[[0], [1], [0]], wO=[[1], [0], [0]],
                   commitmentWeights=[[], [], []], cs=[35, 5, 7])
    witness = dict(assignment=dict(aL=[7], aR=[5], aO=[35]), commitments=[], commitBlinders=[])
    return circuit, witness


def shuffle_circuit(rng):
    """ArithCircuit.hs:167-207: n = 2, m = 4, q = 5."""
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
    return circuit, witness


def arbitrary_circuit(rng, n, m):
    """arithCircuitGen / arithAssignmentGen / arithWitnessGen (ArithmeticCircuit/Internal.hs:241-295):
    q = m constraints, each weight matrix one all-ones row, wV a shuffled identity."""
    def gen_vec():
        i = rng.randrange(0, m)
        return [[1] * n if r == i else [0] * n for r in range(m)]
    w_l, w_r, w_o = gen_vec(), gen_vec(), gen_vec()
    perm = list(range(m))
    rng.shuffle(perm)
    w_v = [[1 if perm[r] == j else 0 for j in range(m)] for r in range(m)]
    cs = [rng.randrange(R) for _ in range(m)]
    a_l = [rng.randrange(0, 2**n + 1) % R for _ in range(n)]
    a_r = [rng.randrange(0, 2**n + 1) % R for _ in range(n)]
    a_o = o.hadamard(a_l, a_r)
    # computeInputValues (:176-183): wV v = wL aL + wR aR + wO aO - cs, wV a permutation
    sol = o.vsub(o.vadd(o.vadd(o.vector_matrix_product_t(a_l, w_l), o.vector_matrix_product_t(a_r, w_r)),
                        o.vector_matrix_product_t(a_o, w_o)), cs)
    vs = [0] * m
    for r in range(m):
        vs[perm[r]] = sol[r]
    bl = [rng.randrange(R) for _ in range(m)]
    circuit = dict(wL=w_l, wR=w_r, wO=w_o, commitmentWeights=w_v, cs=cs)
    witness = dict(assignment=dict(aL=a_l, aR=a_r, aO=a_o),
                   commitments=[o.commit(v, b) for v, b in zip(vs, bl)], commitBlinders=bl)
    return circuit, witness


def mutations(rng, proof, commitments):
    ipp = proof["productProof"]
    out = [(dict(proof, t=(proof["t"] + 1) % R), commitments),
           (dict(proof, mu=(proof["mu"] + 1) % R), commitments),
           (dict(proof, tBlinding=(proof["tBlinding"] + 1) % R), commitments),
           (dict(proof, aoCommit=o.point_add(proof["aoCommit"], o.GEN)), commitments),
           (dict(proof, productProof=dict(ipp, l=(ipp["l"] + 1) % R)), commitments)]
    if commitments:
        out.append((proof, [o.point_add(commitments[0], o.GEN)] + list(commitments[1:])))
    if ipp["lCommits"]:
        out.append((dict(proof, productProof=dict(ipp, lCommits=ipp["rCommits"], rCommits=ipp["lCommits"])),
                    commitments))
    t_c = list(proof["tCommits"])
    t_c[3] = o.point_add(t_c[3], o.GEN)
    out.append((dict(proof, tCommits=t_c), commitments))
    return out


CASES = {
    "hadamard16": lambda rng: hadamard_circuit(rng, 16),
    "hadamard5_padded": lambda rng: hadamard_circuit(rng, 5),
    "no_mult_gates": no_mult_gates,
    "no_input_values": no_input_values,
    "shuffle": shuffle_circuit,
    "arbitrary_n20_m3": lambda rng: arbitrary_circuit(rng, 20, 3),
    "arbitrary_n33_m7": lambda rng: arbitrary_circuit(rng, 33, 7),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_circuit_prove_verify_vs_oracle(eng_gens, name):
    rng = random.Random(sum(map(ord, name)))
    circuit, witness = CASES[name](rng)
    n = len(witness["assignment"]["aL"])
    n2 = len(o.pad_to_nearest_power_of_two([0] * n))
    blinders = o.ac_draw_blinders(rng, n2)
    exp = o.ac_generate_proof(circuit, witness, *blinders)
    got = eng_gens.ac_prove(circuit, witness, *blinders)
    assert got == exp
    assert o.ac_verify_proof(witness["commitments"], exp, circuit)
    assert eng_gens.ac_verify(witness["commitments"], got, circuit) is True
    wants = []
    for bad_proof, bad_commits in mutations(rng, exp, witness["commitments"]):
        want = o.ac_verify_proof(bad_commits, bad_proof, circuit)
        assert eng_gens.ac_verify(bad_commits, bad_proof, circuit) == want
        wants.append(want)
    if n:
        assert not any(wants)
    else:
        # with no gates the reference's final check is (l*r)*u == acc with r = 0, so a changed `l` still
        # verifies (mutation index 4); the engine must agree with that, and reject the rest
        assert wants[4] is True and not any(wants[:4] + wants[5:])


def test_unsatisfied_circuit_is_rejected(eng_gens):
    """A witness that breaks a multiplication gate still yields a proof (the prover does not check);
    both verifiers must reject it."""
    rng = random.Random(99)
    circuit, witness = hadamard_circuit(rng, 8)
    witness["assignment"]["aO"][3] = (witness["assignment"]["aO"][3] + 1) % R
    blinders = o.ac_draw_blinders(rng, 8)
    exp = o.ac_generate_proof(circuit, witness, *blinders)
    got = eng_gens.ac_prove(circuit, witness, *blinders)
    assert got == exp
    assert eng_gens.ac_verify(witness["commitments"], got, circuit) is False
    assert o.ac_verify_proof(witness["commitments"], exp, circuit) is False


def test_config5_full_size_vs_oracle_fixture(engine):
    """BASELINE config 5 at FULL size (n = 2^12 gates, Q = 2) against the oracle: tests/golden/config5_vectors.json
    holds what oracle/bp_oracle.py (ArithmeticCircuit/Prover.hs:16-117, Verifier.hs:16-71) computed for the seed-5
    inputs of util.config5_inputs - the whole proof (34 points + 5 scalars), a SHA-256 over the points, and the verdicts
    of the honest proof and three mutations (tests/golden/make_golden_config5.py; ~7 minutes of Python).  bp_ac_prove
    must return that proof byte for byte and bp_ac_verify those verdicts."""
    import hashlib
    import json
    import os
    from util import HERE, config5_inputs, config5_mutations, ux, up
    with open(os.path.join(HERE, "golden", "config5_vectors.json")) as f:
        g = json.load(f)
    n = g["n"]
    engine.derive_generators(n)
    circuit, witness, openings, blinders = config5_inputs(o, n, g["seed"])
    commitments = engine.add_two_mul_p([v for v, _ in openings], [o.GEN] * 2, [b for _, b in openings], [o.h()] * 2)
    assert commitments == [up(p) for p in g["commitments"]]
    witness = dict(witness, commitments=commitments)
    pj = g["proof"]
    want = dict(tBlinding=ux(pj["tBlinding"]), mu=ux(pj["mu"]), t=ux(pj["t"]), aiCommit=up(pj["aiCommit"]),
                aoCommit=up(pj["aoCommit"]), sCommit=up(pj["sCommit"]), tCommits=[up(p) for p in pj["tCommits"]],
                productProof=dict(lCommits=[up(p) for p in pj["lCommits"]], rCommits=[up(p) for p in pj["rCommits"]],
                                  l=ux(pj["l"]), r=ux(pj["r"])))
    got = engine.ac_prove(circuit, witness, *blinders)
    assert len(got["productProof"]["lCommits"]) == 12
    ipp = got["productProof"]
    pts = [got["aiCommit"], got["aoCommit"], got["sCommit"]] + list(got["tCommits"]) + list(ipp["lCommits"]) \
        + list(ipp["rCommits"])
    assert hashlib.sha256(b"".join(o.wire_point(p) for p in pts)).hexdigest() == g["points_sha256"]
    assert got == want
    # the transcript values the oracle saw on the way (host layer of the engine against the fixture)
    from bulletproofs_amd import _lib
    assert _lib.transcript(0, 7, pts=[got["aiCommit"], got["aoCommit"], got["sCommit"]]) == ux(g["y"])
    assert _lib.transcript(0, 8, scs=[ux(g["y"])]) == ux(g["z"])
    assert _lib.transcript(0, 9, pts=got["tCommits"], raw=b"x" * 7) == ux(g["x"])
    assert engine.mul([o.GEN], [o.shamir_u(want["tBlinding"], want["mu"], want["t"])])[0] == up(g["u"])
    v = g["verdicts"]
    assert engine.ac_verify(commitments, want, circuit) is v["honest"] is True
    verdicts = [engine.ac_verify(bc, bp, circuit) for bp, bc in config5_mutations(o, want, commitments)]
    assert verdicts == [v["t_plus_1"], v["commitments_swapped"], v["l0_r0_swapped"]] == [False, False, False]
    engine.set_generators(o.gs(128), o.hs(128), o.h())


def test_golden_circuit_proofs(eng_gens):
    """The committed fixtures (tests/golden/circuit_vectors.json): engine proof == fixture, verdicts agree."""
    from util import circuit_golden, load_circuit_case
    for name, c in circuit_golden().items():
        circuit, witness, blinders, proof = load_circuit_case(c)
        assert eng_gens.ac_prove(circuit, witness, *blinders) == proof, name
        assert eng_gens.ac_verify(witness["commitments"], proof, circuit) is True
        assert eng_gens.ac_verify(witness["commitments"], dict(proof, mu=(proof["mu"] + 1) % R), circuit) is False
