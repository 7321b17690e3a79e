"""CPU tests of the oracle: pinned against OpenSSL's secp256k1, the committed
golden vectors, and the properties the reference's own test-suite states
(test/Test/Field.hs, test/Test/Protocol/RangeProof.hs)."""
import ctypes
import ctypes.util
import random

import pytest

import bp_oracle as o
from util import golden, ux, up, load_proof, load_blinders


# ---- independent pin: OpenSSL libcrypto --------------------------------------------
class OpenSSL:
    def __init__(self):
        name = ctypes.util.find_library("crypto")
        if not name:
            pytest.skip("libcrypto not present")
        L = ctypes.CDLL(name)
        vp = ctypes.c_void_p
        L.EC_GROUP_new_by_curve_name.restype = vp
        L.EC_POINT_new.restype = vp
        L.EC_POINT_new.argtypes = [vp]
        L.BN_new.restype = vp
        L.BN_bin2bn.restype = vp
        L.BN_bin2bn.argtypes = [ctypes.c_char_p, ctypes.c_int, vp]
        L.BN_CTX_new.restype = vp
        L.EC_POINT_mul.argtypes = [vp] * 6
        L.EC_POINT_add.argtypes = [vp] * 5
        L.EC_POINT_set_affine_coordinates.argtypes = [vp] * 5
        L.EC_POINT_get_affine_coordinates.argtypes = [vp] * 5
        L.EC_POINT_is_at_infinity.argtypes = [vp, vp]
        L.BN_bn2binpad.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
        self.L = L
        self.g = L.EC_GROUP_new_by_curve_name(714)  # NID_secp256k1
        self.ctx = L.BN_CTX_new()

    def bn(self, v):
        return self.L.BN_bin2bn(v.to_bytes(32, "big"), 32, None)

    def pt(self, p):
        q = self.L.EC_POINT_new(self.g)
        assert self.L.EC_POINT_set_affine_coordinates(self.g, q, self.bn(p[0]), self.bn(p[1]), self.ctx) == 1
        return q

    def out(self, q):
        if self.L.EC_POINT_is_at_infinity(self.g, q):
            return None
        x, y = self.L.BN_new(), self.L.BN_new()
        assert self.L.EC_POINT_get_affine_coordinates(self.g, q, x, y, self.ctx) == 1
        bx, by = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        self.L.BN_bn2binpad(x, bx, 32)
        self.L.BN_bn2binpad(y, by, 32)
        return int.from_bytes(bx.raw, "big"), int.from_bytes(by.raw, "big")

    def mul(self, p, k):
        q = self.L.EC_POINT_new(self.g)
        assert self.L.EC_POINT_mul(self.g, q, None, self.pt(p), self.bn(k), self.ctx) == 1
        return self.out(q)

    def add(self, a, b):
        q = self.L.EC_POINT_new(self.g)
        assert self.L.EC_POINT_add(self.g, q, self.pt(a), self.pt(b), self.ctx) == 1
        return self.out(q)


@pytest.fixture(scope="module")
def ossl():
    return OpenSSL()


def test_point_mul_vs_openssl(ossl):
    rng = random.Random(11)
    base = o.GEN
    for _ in range(25):
        k = rng.randrange(1, o.R)
        assert o.point_mul(base, k) == ossl.mul(base, k)
        base = o.point_mul(o.GEN, rng.randrange(1, o.R))
    assert o.point_mul(o.GEN, o.R) is None


def test_point_add_vs_openssl(ossl):
    rng = random.Random(12)
    for _ in range(25):
        a = o.point_mul(o.GEN, rng.randrange(1, o.R))
        b = o.point_mul(o.GEN, rng.randrange(1, o.R))
        assert o.point_add(a, b) == ossl.add(a, b)
        assert o.point_add(a, a) == ossl.add(a, a)
        assert o.point_add(a, o.point_neg(a)) is None


def test_known_kg_constants():
    # 2G and 3G of secp256k1 (widely published)
    assert o.point_mul(o.GEN, 2) == (
        0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
        0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    assert o.point_mul(o.GEN, 3) == (
        0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
        0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672)


# ---- golden vectors ---------------------------------------------------------------------
def test_golden_generators_and_points():
    g = golden()
    assert up(g["h"]) == o.h()
    assert [up(p) for p in g["gs"]] == o.gs(4)
    assert [up(p) for p in g["hs"]] == o.hs(4)
    for e in g["kG"]:
        assert o.point_mul(o.GEN, ux(e["k"])) == up(e["P"])
    for c in g["sumExps"]:
        assert o.sum_exps([ux(s) for s in c["scalars"]], [up(p) for p in c["points"]]) == up(c["result"])


def test_golden_bench_proof():
    """The reference's bench input (bench/Bench/RangeProof.hs:10-14)."""
    b = golden()["bench_proof"]
    pr = o.mrp_generate_proof_unsafe(2**64, [(b["v"], b["vBlinding"])], *load_blinders(b["blinders"]))
    assert pr == load_proof(b["proof"])
    assert o.commit(b["v"], b["vBlinding"]) == up(b["vCommit"])
    assert o.rp_verify_proof(2**64, up(b["vCommit"]), pr) is True


def test_golden_multi_proof():
    m = golden()["multi_proof"]
    vs = [(ux(v), ux(b)) for v, b in m["vs"]]
    pr = o.mrp_generate_proof(2**8, vs, *load_blinders(m["blinders"]))
    assert pr == load_proof(m["proof"])
    assert o.mrp_verify_proof(2**8, [up(p) for p in m["vCommits"]], pr) is True


# ---- properties restated from the reference's tests ------------------------------------------
def test_prop_add_sub_mod():
    """test/Test/Field.hs:15-33."""
    rng = random.Random(13)
    for _ in range(5):
        x, y = rng.randrange(o.R), rng.randrange(o.R)
        assert o.point_mul(o.GEN, (x + y) % o.R) == o.point_add(o.point_mul(o.GEN, x), o.point_mul(o.GEN, y))
        assert o.point_mul(o.GEN, (x - y) % o.R) == o.point_add(o.point_mul(o.GEN, x),
                                                                 o.point_neg(o.point_mul(o.GEN, y)))


def test_sum_exps_truncation_and_empty():
    """Utils.hs:79-80."""
    p = o.gs(3)
    assert o.sum_exps([], p) is None
    assert o.sum_exps([5], []) is None
    assert o.sum_exps([5, 7, 9], p[:2]) == o.sum_exps([5, 7], p[:2])
    assert o.sum_exps([5, 7], p) == o.sum_exps([5, 7], p[:2])


def test_power_vector_zero_case():
    """Utils.hs:50-52: 0^0 = 0."""
    assert o.power_vector(0, 3) == [0, 0, 0]
    assert o.power_vector(2, 4) == [1, 2, 4, 8]
    assert o.power_vector(1, 3) == [1, 1, 1]


def test_bit_vector_algebra():
    """test/Test/Protocol/RangeProof.hs:29-103."""
    rng = random.Random(14)
    for _ in range(6):
        n = 2 ** rng.randrange(0, 8)
        v = rng.randrange(1, 2**n)
        a_l = o.reversed_encode_bit(n, v)
        a_r = o.complementary_vector(a_l)
        assert o.dot(a_l, a_r) == 0
        assert o.hadamard(a_l, a_r) == [0] * n
        assert o.dot(a_l, o.power_vector(2, n)) == v % o.R
        y, z = rng.randrange(o.R), rng.randrange(o.R)
        yn = o.power_vector(y, n)
        assert o.dot(o.vsub(o.vsub(a_l, o.power_vector(1, n)), a_r), yn) == 0
        # obfuscateEncodedBitsSingle == z^2 v + delta n 1 y z
        z1n = [z] * n
        lhs = o.dot(o.vsub(a_l, z1n),
                    o.vadd(o.hadamard(yn, o.vadd(a_r, z1n)), [z * z * t % o.R for t in o.power_vector(2, n)]))
        assert lhs == (z * z * v + o.delta(n, 1, y, z)) % o.R


@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (4, 1), (8, 3), (16, 1)])
def test_completeness_and_soundness(n, m):
    """test/Test/Protocol/RangeProof.hs:112-145,183-221 (small sizes; the
    reference samples n in {1..128}, m in 1..10)."""
    rng = random.Random(100 * n + m)
    ub = 2**n
    vs = [(rng.randrange(0, 2**n), rng.randrange(1, o.R)) for _ in range(m)]
    m2 = 2 ** o.log2_ceil(m)
    pr = o.mrp_generate_proof(ub, vs, *o.draw_blinders(rng, n * m2))
    vcs = [o.commit(v, b) for v, b in vs]
    y = o.shamir_y(pr["aCommit"], pr["sCommit"])
    z = o.shamir_z(pr["aCommit"], pr["sCommit"], y)
    x = o.shamir_x(pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"], y, z)
    padded = vcs + [None] * (m2 - m)
    assert o.verify_t_poly(n, padded, pr, x, y, z)
    assert o.verify_lr_commitment(n, m2, pr, x, y, z)
    assert o.mrp_verify_proof(ub, vcs, pr)
    bad = [o.commit(vs[0][0] + 1, vs[0][1])] + vcs[1:]
    assert not o.mrp_verify_proof(ub, bad, pr)
    if n > 1:
        assert not o.mrp_verify_proof(2 ** (n // 2), vcs, pr)  # different upper bound


def test_error_paths():
    """test/Test/Protocol/RangeProof.hs:147-171."""
    rng = random.Random(15)
    bl = o.draw_blinders(rng, 8)
    with pytest.raises(o.RangeProofError) as e:
        o.mrp_generate_proof(2**8, [(300, 5)], *bl)
    assert e.value.args[0][0] == "ValuesNotInRange"
    with pytest.raises(o.RangeProofError) as e:
        o.mrp_generate_proof(o.R + 1, [(3, 5)], *bl)
    assert e.value.args[0][0] == "UpperBoundTooLarge"
    with pytest.raises(o.RangeProofError) as e:
        o.mrp_generate_proof(2**6, [(3, 5)], *bl)  # n = 6 is not a power of two
    assert e.value.args[0][0] == "NNotPowerOf2"


def test_ipp_standalone():
    """InnerProductProof generateProof/verifyProof on an arbitrary base."""
    rng = random.Random(16)
    for n in [1, 2, 8]:
        g, hh = o.gs(n), o.hs(n)
        u = o.point_mul(o.GEN, rng.randrange(1, o.R))
        ls = [rng.randrange(o.R) for _ in range(n)]
        rs = [rng.randrange(o.R) for _ in range(n)]
        c = o.point_add(o.point_add(o.sum_exps(ls, g), o.sum_exps(rs, hh)), o.point_mul(u, o.dot(ls, rs)))
        pr = o.ipp_generate_proof(g, hh, u, c, ls, rs)
        assert len(pr["lCommits"]) == n.bit_length() - 1
        assert o.ipp_verify_proof(n, g, hh, u, c, pr)
        pr2 = dict(pr, l=(pr["l"] + 1) % o.R)
        assert not o.ipp_verify_proof(n, g, hh, u, c, pr2)
    assert o.ipp_generate_proof([], [], u, None, [], []) == dict(lCommits=[], rCommits=[], l=0, r=0)


def test_log_base2_matches_double_arithmetic():
    assert all(o.log_base2(2**k) == k for k in range(1, 257))
    assert o.log_base2(2**64 + 1) == 64 and o.log_base2(2**64 - 1) == 64  # Double rounding, as in Haskell


# ---- arithmetic circuits (ArithmeticCircuit/*.hs) ----------------------------------------------
def test_golden_circuit_proofs():
    from util import circuit_golden, load_circuit_case
    for name, c in circuit_golden().items():
        circuit, witness, blinders, proof = load_circuit_case(c)
        assert o.ac_generate_proof(circuit, witness, *blinders) == proof, name
        assert o.ac_verify_proof(witness["commitments"], proof, circuit) is True
        assert o.ac_verify_proof(witness["commitments"], dict(proof, mu=(proof["mu"] + 1) % o.R), circuit) is False


def test_circuit_completeness_on_reference_circuits():
    """test/Test/Protocol/ArithCircuit.hs: n = 0 / m = 3 (:92-117) and n = 1 / m = 0 (:130-153)."""
    rng = random.Random(77)
    bl = [rng.randrange(o.R) for _ in range(3)]
    circuit = dict(wL=[[]], wR=[[]], wO=[[]], commitmentWeights=[[1, 1, -1]], cs=[0])
    witness = dict(assignment=dict(aL=[], aR=[], aO=[]), commitments=[o.commit(v, b) for v, b in zip([2, 5, 7], bl)],
                   commitBlinders=bl)
    pr = o.ac_generate_proof(circuit, witness, *o.ac_draw_blinders(rng, 0))
    assert o.ac_verify_proof(witness["commitments"], pr, circuit)
    bad = [o.commit(3, bl[0])] + witness["commitments"][1:]
    assert not o.ac_verify_proof(bad, pr, circuit)
    circuit = dict(wL=[[0], [0], [1]], wR=[[0], [1], [0]], wO=[[1], [0], [0]], commitmentWeights=[[], [], []],
                   cs=[35, 5, 7])
    witness = dict(assignment=dict(aL=[7], aR=[5], aO=[35]), commitments=[], commitBlinders=[])
    pr = o.ac_generate_proof(circuit, witness, *o.ac_draw_blinders(rng, 1))
    assert o.ac_verify_proof([], pr, circuit)
    assert not o.ac_verify_proof([], pr, dict(circuit, cs=[36, 5, 7]))


def test_config5_fixture_is_consistent():
    """tests/golden/config5_vectors.json (config 5 at full size, written by make_golden_config5.py): the stored
    challenges are the oracle's transcript of the stored proof points, the digest is over those points, and the inputs
    the GPU test regenerates from the seed have the stored commitments' openings (cheap checks only: the full
    prove / verify takes the oracle minutes and is what the generator script ran)."""
    import hashlib
    import json
    import os
    from util import HERE, config5_inputs, ux, up
    with open(os.path.join(HERE, "golden", "config5_vectors.json")) as f:
        g = json.load(f)
    pj = g["proof"]
    ai, ao, s = up(pj["aiCommit"]), up(pj["aoCommit"]), up(pj["sCommit"])
    tcs = [up(p) for p in pj["tCommits"]]
    ls, rs = [up(p) for p in pj["lCommits"]], [up(p) for p in pj["rCommits"]]
    assert len(ls) == len(rs) == 12 and len(tcs) == 7
    y = o.shamir_gxgxg(ai, ao, s)
    assert y == ux(g["y"]) and o.ac_shamir_z(y) == ux(g["z"]) and o.shamir_gs(tcs) == ux(g["x"])
    assert o.point_mul(o.GEN, o.shamir_u(ux(pj["tBlinding"]), ux(pj["mu"]), ux(pj["t"]))) == up(g["u"])
    pts = [ai, ao, s] + tcs + ls + rs
    assert all(o.is_on_curve(p) for p in pts)
    assert hashlib.sha256(b"".join(o.wire_point(p) for p in pts)).hexdigest() == g["points_sha256"]
    _, _, openings, _ = config5_inputs(o, g["n"], g["seed"])
    assert [o.commit(v, b) for v, b in openings] == [up(p) for p in g["commitments"]]
    assert g["verdicts"] == dict(honest=True, t_plus_1=False, commitments_swapped=False, l0_r0_swapped=False)


def test_verifier_is_total_over_bounds_and_pair_counts():
    """MultiRangeProof/Verifier.hs:25-39 checks neither the bound nor the proof's shape: the oracle follows it for
    bounds that are no power of two, n = 0 (and upperBound = 0, whose logBase2 GHC floors to -2^1024), surplus L/R
    pairs, and raises where `challenges !! j` (InnerProductProof/Verifier.hs:72) or `2 ^ log2Ceil 0` would."""
    rng = random.Random(41)
    vs = [(rng.randrange(0, 2**8), rng.randrange(1, o.R))]
    pr = o.mrp_generate_proof(2**8, vs, *o.draw_blinders(rng, 8))
    vcs = [o.commit(*vs[0])]
    assert o.log_base2(0) == -(2**1024) and o.log_base2(12) == 3 and o.log_base2(2**64 - 1) == 64
    for ub in (2**4, 2**16, 12, 1, 0, 2**7, 3):          # prop_differentUpperBound: verifyTPoly fails first
        assert o.mrp_verify_proof(ub, vcs, pr) is False
    assert o.mrp_verify_proof(2**8 + 1, vcs, pr) is True  # logBase2 floors: still n = 8
    y = o.shamir_y(pr["aCommit"], pr["sCommit"])
    z = o.shamir_z(pr["aCommit"], pr["sCommit"], y)
    for ub, want in ((12, False), (2**4, False), (1, False), (0, False), (2**16, IndexError)):
        n2 = o.log_base2(ub)
        f = dict(pr, t=(pr["t"] - o.delta(8, 1, y, z) + o.delta(n2, 1, y, z)) % o.R)  # verifyTPoly holds under ub
        if want is IndexError:
            with pytest.raises(IndexError):
                o.mrp_verify_proof(ub, vcs, f)
        else:
            assert o.mrp_verify_proof(ub, vcs, f) is want
    ipp = pr["productProof"]
    longer = dict(pr, productProof=dict(ipp, lCommits=ipp["lCommits"] + [o.GEN], rCommits=ipp["rCommits"] + [o.GEN]))
    assert o.mrp_verify_proof(2**8, vcs, longer) is False
    with pytest.raises(IndexError):
        o.mrp_verify_proof(2**8, vcs, dict(pr, productProof=dict(ipp, lCommits=ipp["lCommits"][:-1])))
    with pytest.raises(ArithmeticError):
        o.mrp_verify_proof(2**8, [], pr)
    # n m = 0: mkOtherExponents still holds key 0 and bGs = gs is infinite, so gsCommit = s0 * gs!!0
    assert o.mk_other_exponents(0, [5, 7]) == [pow(35, -1, o.R)]
    assert o.mk_other_exponents(12, [2, 3, 5])[:3] == [pow(30, -1, o.R), pow(30, -1, o.R) * 4 % o.R,
                                                      pow(30, -1, o.R) * 9 % o.R]
    assert len(o.mk_other_exponents(12, [2, 3, 5])) == 10  # keys 10, 11 are never inserted: sumExps truncates
