"""CPU tests of the device arithmetic headers (fe.h, ec.h, msm_core.h) compiled
for the host (tests/hostsim), against the oracle and the golden vectors."""
import ctypes
import os
import random
import subprocess

import pytest

import bp_oracle as o
from util import golden, ux, up

HERE = os.path.dirname(os.path.abspath(__file__))
P, R = o.P, o.R


@pytest.fixture(scope="module")
def hs():
    so = os.path.join(HERE, "hostsim", "libhostsim.so")
    src = os.path.join(HERE, "hostsim", "hostsim.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "bulletproofs_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)
                                            if f.endswith(".h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


def b(x):
    return int(x).to_bytes(32, "little")


def pb(pt):
    return bytes(64) if pt is None else b(pt[0]) + b(pt[1])


def dp(raw):
    x, y = int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)


def fq(hs, op, a, c=0):
    out = ctypes.create_string_buffer(32)
    hs.hs_fq_op(op, b(a), b(c), out)
    return int.from_bytes(out.raw, "little")


def mont(hs, which, op, a, c=0):
    out = ctypes.create_string_buffer(32)
    hs.hs_mont_op(which, op, b(a), b(c), out)
    return int.from_bytes(out.raw, "little")


def test_fq_weak_representation_edges(hs):
    rng = random.Random(7)
    edge = [0, 1, 2, P - 1, P, P + 1, 2**256 - 1, 2**256 - 2, 2**32 + 977, 2**32 + 976, P - 2**32, 2**255, 977]
    vals = edge + [rng.randrange(2**256) for _ in range(120)]
    for a in vals:
        for c in edge + [rng.choice(vals) for _ in range(6)]:
            assert fq(hs, 0, a, c) == (a + c) % P
            assert fq(hs, 1, a, c) == (a - c) % P
            assert fq(hs, 2, a, c) == (a * c) % P
        assert fq(hs, 3, a) == a * a % P
        assert fq(hs, 6, a) == (-a) % P
        assert bool(hs.hs_fq_is_zero(b(a))) == (a % P == 0)
    for a in vals[:40]:
        assert fq(hs, 4, a) == (pow(a, -1, P) if a % P else 0)
        if pow(a % P, (P - 1) // 2, P) == 1:
            assert fq(hs, 5, a) ** 2 % P == a % P


@pytest.mark.parametrize("which,mod", [(0, R), (1, P)])
def test_montgomery_field_laws(hs, which, mod):
    """test/Test/Field.hs:39-64 field laws + exact values."""
    rng = random.Random(8 + which)
    ev = [0, 1, 2, mod - 1, mod - 2] + [rng.randrange(mod) for _ in range(60)]
    for a in ev:
        for c in ev[:6] + [rng.choice(ev) for _ in range(4)]:
            assert mont(hs, which, 0, a, c) == (a + c) % mod
            assert mont(hs, which, 1, a, c) == (a - c) % mod
            assert mont(hs, which, 2, a, c) == (a * c) % mod
        assert mont(hs, which, 3, a) == a * a % mod
        assert mont(hs, which, 5, a) == (-a) % mod
    for a in ev[:20]:
        assert mont(hs, which, 4, a) == (pow(a, -1, mod) if a else 0)


def test_golden_field(hs):
    g = golden()
    for e in g["fq"]:
        a, c = ux(e["a"]), ux(e["b"])
        assert fq(hs, 0, a, c) == ux(e["add"]) and fq(hs, 1, a, c) == ux(e["sub"])
        assert fq(hs, 2, a, c) == ux(e["mul"]) and fq(hs, 4, a) == ux(e["inv"])
    for e in g["fr"]:
        a, c = ux(e["a"]), ux(e["b"])
        assert mont(hs, 0, 0, a, c) == ux(e["add"]) and mont(hs, 0, 1, a, c) == ux(e["sub"])
        assert mont(hs, 0, 2, a, c) == ux(e["mul"]) and mont(hs, 0, 4, a) == ux(e["inv"])


def pop(hs, op, p, q=None, k=0):
    out = ctypes.create_string_buffer(64)
    hs.hs_point_op(op, pb(p), pb(q), b(k), out)
    return dp(out.raw)


def test_group_law_total(hs):
    rng = random.Random(9)
    G = o.GEN
    pts = [None, G, o.point_neg(G)] + [o.point_mul(G, rng.randrange(1, R)) for _ in range(8)]
    for p in pts:
        assert pop(hs, 2, p) == o.point_add(p, p)
        assert pop(hs, 3, p) == o.point_neg(p)
        assert pop(hs, 0, p, o.point_neg(p)) is None
        for q in pts:
            assert pop(hs, 0, p, q) == o.point_add(p, q)
            assert pop(hs, 1, p, q) == o.point_add(p, q)
    for k in [0, 1, 2, R - 1, R - 2] + [rng.randrange(R) for _ in range(6)]:
        for p in pts[:4]:
            assert pop(hs, 4, p, None, k) == o.point_mul(p, k)
    for e in golden()["kG"]:
        assert pop(hs, 4, G, None, ux(e["k"])) == up(e["P"])
    assert hs.hs_on_curve(pb(G)) == 1 and hs.hs_on_curve(pb((5, 7))) == 0 and hs.hs_on_curve(pb(None)) == 1


def test_device_decimal_records(hs):
    """dec.h (the formatter the kernels use for the transcript's `show` text): the 80-byte record of a 256-bit
    integer carries str(v) right-aligned and its length, for edge values and random ones; a point shown from
    two records equals the host formatter's text in both `show` styles."""
    rng = random.Random(123)
    vals = [0, 1, 9, 10, 10**9 - 1, 10**9, 10**18, 2**32 - 1, 2**32, 2**64, 10**77 - 1, 10**77, P - 1, P, R, 2**256 - 1]
    vals += [10**e for e in range(0, 78)] + [10**e - 1 for e in range(1, 78)]
    vals += [rng.randrange(2**rng.randrange(1, 257)) for _ in range(20000)]
    rec = ctypes.create_string_buffer(80)
    for v in vals:
        hs.hs_dec_record(b(v), rec)
        n = rec.raw[0]
        assert rec.raw[80 - n:80].decode() == str(v), v
    out1, out2 = ctypes.create_string_buffer(256), ctypes.create_string_buffer(256)
    hs.hs_show_point_rec.restype = ctypes.c_size_t
    hs.hs_show_point.restype = ctypes.c_size_t
    pts = [None, o.GEN, (1, 0), (0, 1)] + [(rng.randrange(P), rng.randrange(P)) for _ in range(200)]
    for style in (0, 1):
        for pt in pts:
            raw = pb(pt) if pt is not None else bytes(64)
            rx, ry = ctypes.create_string_buffer(80), ctypes.create_string_buffer(80)
            hs.hs_dec_record(raw[:32], rx)
            hs.hs_dec_record(raw[32:], ry)
            n1 = hs.hs_show_point_rec(style, rx.raw + ry.raw, out1)
            n2 = hs.hs_show_point(style, raw, out2)
            assert out1.raw[:n1] == out2.raw[:n2]


def test_device_transcript_matches_the_oracle(hs):
    """fsdev.h (the challenges formed on the GPU: verify.hip k_fs_initial / k_fs_round): `show` text, SHA-256 and the
    digest as a scalar equal the oracle's shamirY / Z / X / U / X' (Utils.hs:145-174) in both `show` styles, for
    random and edge inputs (O, small coordinates with short decimal text, digests >= r are covered by the reduction
    test below), with the message buffer contiguous or strided as on the device."""
    rng = random.Random(91)
    G = o.GEN
    pts = [o.point_mul(G, rng.randrange(1, R)) for _ in range(6)] + [None, (1, 2), (0, 5), (10**9, 10**18), (P - 1, P - 1)]
    scs = [0, 1, 9, 10, 10**76, 10**77, R - 1, R - 2, 2**255] + [rng.randrange(R) for _ in range(6)]
    old = o.SHOW_STYLE
    try:
        for style, name in ((0, "derived"), (1, "plain")):
            o.SHOW_STYLE = name
            for it in range(40):
                p4 = [rng.choice(pts) for _ in range(4)]
                s3 = [rng.choice(scs) for _ in range(3)]
                stride = rng.choice([1, 1, 3, 64])
                exp = [o.shamir_y(p4[0], p4[1]), o.shamir_z(p4[0], p4[1], s3[0]),
                       o.shamir_x(p4[0], p4[1], p4[2], p4[3], s3[0], s3[1]), o.shamir_u(s3[0], s3[1], s3[2]),
                       o.shamir_x_prime(p4[2], p4[0], p4[1])]
                for which in range(5):
                    out = ctypes.create_string_buffer(32)
                    hs.hs_fs_challenge(style, which, b"".join(pb(q) for q in p4), b"".join(b(x) for x in s3),
                                       ctypes.c_size_t(stride), out)
                    assert int.from_bytes(out.raw, "little") == exp[which], (style, which, it)
                out3 = ctypes.create_string_buffer(96)
                hs.hs_fs_yzx(style, b"".join(pb(q) for q in p4), ctypes.c_size_t(stride), out3)
                y = o.shamir_y(p4[0], p4[1])
                z = o.shamir_z(p4[0], p4[1], y)
                x = o.shamir_x(p4[0], p4[1], p4[2], p4[3], y, z)
                assert [int.from_bytes(out3.raw[32 * i:32 * i + 32], "little") for i in range(3)] == [y, z, x]
                out = ctypes.create_string_buffer(32)
                hs.hs_fs_u(style, b"".join(b(v) for v in s3), ctypes.c_size_t(stride), out)
                assert int.from_bytes(out.raw, "little") == o.shamir_u(s3[0], s3[1], s3[2])
                # the round's challenge from a hashed opening (show _r <> L <> R) continued with the commitment
                hs.hs_fs_round(style, b"".join(pb(q) for q in p4[:3]), ctypes.c_size_t(stride),
                               ctypes.c_size_t(rng.choice([1, 5, 64])), out)
                assert int.from_bytes(out.raw, "little") == o.shamir_x_prime(p4[2], p4[0], p4[1])
    finally:
        o.SHOW_STYLE = old


LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def test_glv_positive_split_and_pair_table(hs):
    """glv.h: k = m1 + m2*lambda (mod r) with 0 <= m1, m2 < 2^129 (no signs, so the joint table of a pair of
    points is independent of the scalars), and k0*P0 + k1*P1 through that table equals the oracle's
    addTwoMulP, including identity points, equal and opposite points."""
    rng = random.Random(77)
    ks = [0, 1, 2, R - 1, R - 2, LAMBDA, R - LAMBDA, (R - 1) // 2, (R + 1) // 2, 2**128, 2**255] + \
         [rng.randrange(R) for _ in range(3000)]
    for k in ks:
        o1, o2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        hs.hs_glv_split_pos(b(k), o1, o2)
        m1, m2 = int.from_bytes(o1.raw, "little"), int.from_bytes(o2.raw, "little")
        assert m1 < 2**129 and m2 < 2**129
        assert (m1 + m2 * LAMBDA - k) % R == 0
    G = o.GEN
    A = o.point_mul(G, rng.randrange(1, R))
    C = o.point_mul(G, rng.randrange(1, R))
    beta_a = o.point_mul(A, LAMBDA)
    pairs = [(A, C), (A, A), (A, o.point_neg(A)), (None, C), (A, None), (None, None), (A, beta_a),
             (A, o.point_neg(beta_a))]
    for p0, p1 in pairs:
        for k0, k1 in [(rng.randrange(R), rng.randrange(R)), (0, 0), (1, R - 1), (R - 1, R - 1), (5, 0)]:
            out = ctypes.create_string_buffer(64)
            hs.hs_glv_pair_tab(b(k0) + b(k1), pb(p0) + pb(p1), out)
            assert dp(out.raw) == o.add_two_mul_p(k0, p0, k1, p1)
            hs.hs_glv_pair_tab29(b(k0) + b(k1), pb(p0) + pb(p1), out)     # the 29-bit-limb chain of k_round
            assert dp(out.raw) == o.add_two_mul_p(k0, p0, k1, p1)
    for n in (1, 2, 5, 130):
        out = ctypes.create_string_buffer(64)
        hs.hs_dbl29_chain(pb(A), n, out)
        assert dp(out.raw) == o.point_mul(A, 2**n % R)
    hs.hs_dbl29_chain(pb(None), 3, out)
    assert dp(out.raw) is None


def test_two_joint_tables_on_one_chain(hs):
    """glv.h glv_tables_multiexp29 (verify.hip k_stage1_tail, k_var_terms): one or two sign-free joint tables on one
    doubling chain, the second point of a table absent (a single-point table has the entries 1..3 only), against
    the oracle's sum; identity, equal and opposite points included."""
    rng = random.Random(83)
    G = o.GEN
    A, C, D, E = (o.point_mul(G, rng.randrange(1, R)) for _ in range(4))
    edge = [0, 1, 2, R - 1, LAMBDA, R - LAMBDA, 2**128, 2**255]
    cases = []
    for pts in ([A, C, D, E], [A, A, o.point_neg(A), A], [None, C, D, None], [A, o.point_mul(A, LAMBDA), C, C]):
        for present in ([1, 1, 1, 1], [1, 1, 1, 0], [1, 0, 1, 0], [1, 0, 1, 1]):
            for _ in range(3):
                cases.append((2, [rng.choice(edge + [rng.randrange(R)] * 4) for _ in range(4)], present, pts))
        for present in ([1, 0], [1, 1]):
            for _ in range(3):
                cases.append((1, [rng.choice(edge + [rng.randrange(R)] * 4) for _ in range(2)], present, pts[:2]))
    for q in (A, C, o.point_neg(D), None, G):          # P + phi(P) = (beta^2 x, -y), the tables' free composite entry
        out = ctypes.create_string_buffer(64)
        hs.hs_glv_p_plus_phi(pb(q), out)
        assert dp(out.raw) == (o.point_add(q, o.point_mul(q, LAMBDA)) if q else None)
    for nt, ks, present, pts in cases:
        out = ctypes.create_string_buffer(64)
        hs.hs_glv_tables29(nt, b"".join(b(k) for k in ks), bytes(present), b"".join(pb(q) for q in pts), out)
        exp = None
        for k, pr, q in zip(ks, present, pts):
            if pr:
                exp = o.point_add(exp, o.point_mul(q, k))
        assert dp(out.raw) == exp, (nt, ks, present)


def test_glv_bit_lanes(hs):
    """glv.h glv_bit_lane (the verifier's latency path, verify.hip k_round_bits / k_tcheck_bits): the 128 per-bit terms
    on the table 2^j P add up to k*P, for edge scalars, the identity and both signs of the halves."""
    rng = random.Random(78)
    A = o.point_mul(o.GEN, rng.randrange(1, R))
    ks = [0, 1, 2, R - 1, R - 2, LAMBDA, R - LAMBDA, (R - 1) // 2, (R + 1) // 2, 2**127, 2**128, 2**255] + \
         [rng.randrange(R) for _ in range(40)]
    for k in ks:
        out = ctypes.create_string_buffer(64)
        hs.hs_glv_bit_lanes(b(k), pb(A), out)
        assert dp(out.raw) == o.point_mul(A, k)
    out = ctypes.create_string_buffer(64)
    hs.hs_glv_bit_lanes(b(rng.randrange(R)), pb(None), out)
    assert dp(out.raw) is None


def test_fe29_field_ops(hs):
    """fq29.h: nine 29-bit limbs with lazy additions; products, squares, two-product reductions and lazy operands at
    the documented limb bounds against Python integers, edge values included."""
    rng = random.Random(80)
    edge = [0, 1, 2, P - 1, P - 2, 2**255, 2**256 % P, (1 << 29) - 1, ((1 << 261) - 1) % P, 0x1FFFFFFF << 174,
            sum(0x1FFFFFFF << (29 * i) for i in range(0, 9, 2)) % P]
    vals = edge + [rng.randrange(P) for _ in range(200)]
    out = ctypes.create_string_buffer(32)
    for i, a in enumerate(vals):
        c2, d2 = vals[(i * 7 + 3) % len(vals)], vals[(i * 11 + 5) % len(vals)]
        for bb in (vals[(i * 5 + 1) % len(vals)], P - 1, 0):
            hs.hs_fe29_op(0, b(a), b(bb), b(0), b(0), out)
            assert int.from_bytes(out.raw, "little") == a * bb % P
            hs.hs_fe29_op(2, b(a), b(bb), b(c2), b(d2), out)
            assert int.from_bytes(out.raw, "little") == (a * bb + c2 * d2) % P
            out2 = ctypes.create_string_buffer(32)
            hs.hs_fe29_lazy(b(a), b(bb), b(c2), out, out2)
            assert int.from_bytes(out.raw, "little") == ((a - c2) * (bb - c2) - a * bb) % P
            assert int.from_bytes(out2.raw, "little") == ((a - bb) ** 2 - bb - 2 * c2) % P
        hs.hs_fe29_op(1, b(a), b(0), b(0), b(0), out)
        assert int.from_bytes(out.raw, "little") == a * a % P
        hs.hs_fe29_op(3, b(a), b(0), b(0), b(0), out)
        assert int.from_bytes(out.raw, "little") == a
        hs.hs_fe29_op(4, b(a), b(0), b(0), b(0), out)        # Fermat inversion on 29-bit limbs; 0 -> 0
        assert int.from_bytes(out.raw, "little") == (pow(a, -1, P) if a % P else 0)


def test_madd29_chain_is_the_group_sum(hs):
    """xyzz29_madd (fq29.h): sums of table-like points from the identity equal the oracle's sum, through the cases the
    lazy formula cannot do itself and hands to ec.h: identity accumulator, P + P, P + (-P), then going on from O."""
    rng = random.Random(81)
    G = o.GEN
    pts = [o.point_mul(G, rng.randrange(1, R)) for _ in range(12)]

    def run(seq):
        raw = b"".join(pb(p) for p, _ in seq)
        neg = bytes(1 if n else 0 for _, n in seq)
        out = ctypes.create_string_buffer(64)
        hs.hs_madd29_chain(raw, neg, len(seq), out)
        exp = None
        for p, n in seq:
            exp = o.point_add(exp, o.point_neg(p) if n else p)
        assert dp(out.raw) == exp

    run([(p, False) for p in pts])
    run([(p, i % 3 == 0) for i, p in enumerate(pts)])
    A, B2 = pts[0], pts[1]
    run([(A, False), (A, False)])                       # doubling
    run([(A, False), (A, True)])                        # to the identity
    run([(A, False), (A, True), (B2, False)])           # and on from it
    run([(A, False), (A, False), (A, False), (A, True), (A, True), (A, True), (B2, True)])
    run([(A, True), (B2, False), (o.point_add(o.point_neg(A), B2), True), (pts[2], False)])
    for _ in range(20):
        seq = [(rng.choice(pts[:4]), rng.random() < 0.5) for _ in range(rng.randrange(1, 40))]
        run(seq)


def test_add29_full_addition(hs):
    """xyzz29_add (fq29.h): full additions of projective points, with equal, opposite and identity operands."""
    rng = random.Random(82)
    pts = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(6)]
    A = pts[0]
    seqs = [pts, [A, A], [A, o.point_neg(A)], [A, o.point_neg(A), pts[1]], [None, A, None, A, pts[2]], [A, A, A, o.point_neg(A)]]
    for seq in seqs:
        out = ctypes.create_string_buffer(64)
        hs.hs_add29_chain(b"".join(pb(p) for p in seq), len(seq), out)
        exp = None
        for p in seq:
            exp = o.point_add(exp, o.point_mul(p, 4) if p else None)
        assert dp(out.raw) == exp


def test_host_side_inversions(hs):
    """host_inv.h / host_fr.h: the binary extended Euclid the verifiers' latency paths run on the CPU (x_r^-1 mod r,
    1/ZZZ mod p), against Python's modular inverse; 0 maps to 0, weakly reduced base-field inputs are accepted."""
    rng = random.Random(79)
    for which, mod in ((0, P), (1, R), (2, R)):
        vals = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, 2**128, 2**255 % mod] + [rng.randrange(1, mod) for _ in range(300)]
        for a in vals:
            out = ctypes.create_string_buffer(32)
            hs.hs_host_inv(which, b(a), out)
            assert int.from_bytes(out.raw, "little") == pow(a, -1, mod), (which, a)
        out = ctypes.create_string_buffer(32)
        hs.hs_host_inv(which, b(0), out)
        assert out.raw == bytes(32)
    out = ctypes.create_string_buffer(32)
    hs.hs_host_inv(0, b(P + 5), out)      # weak representation of 5
    assert int.from_bytes(out.raw, "little") == pow(5, -1, P)
    # normalisation of an XYZZ point: (x z^2, y z^3, z^2, z^3) -> (x, y)
    A = o.point_mul(o.GEN, rng.randrange(1, R))
    for z in (1, 2, P - 1, rng.randrange(1, P)):
        raw = b(A[0] * z * z % P) + b(A[1] * pow(z, 3, P) % P) + b(z * z % P) + b(pow(z, 3, P))
        out = ctypes.create_string_buffer(64)
        hs.hs_xyzz_to_affine_host(raw, out)
        assert dp(out.raw) == A
    out = ctypes.create_string_buffer(64)
    hs.hs_xyzz_to_affine_host(bytes(128), out)
    assert dp(out.raw) is None


def test_safegcd_inversion(hs):
    """modinv30.h (the per-lane inversions of the verifier's rounds: x_r^-1 mod r, 1/ZZZ mod p): 600 division steps on
    nine signed 30-bit limbs against Python's modular inverse, for both moduli, edge values, weakly reduced
    base-field inputs and the Montgomery-form wrapper; 0 maps to 0."""
    rng = random.Random(84)
    for which, mod in ((0, P), (1, R), (2, R)):
        vals = [1, 2, 3, 4, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, 2**30, 2**30 - 1, 2**60, 2**128, 2**255 % mod,
                2**240, (1 << 256) % mod, mod >> 1, 0x3FFFFFFF << 30, 3**161 % mod]
        vals += [rng.randrange(1, mod) for _ in range(3000)]
        vals += [rng.randrange(1, 2**rng.randrange(1, 256)) for _ in range(500)]
        for a in vals:
            out = ctypes.create_string_buffer(32)
            hs.hs_modinv30(which, b(a), out)
            assert int.from_bytes(out.raw, "little") == pow(a, -1, mod), (which, a)
        out = ctypes.create_string_buffer(32)
        hs.hs_modinv30(which, b(0), out)
        assert out.raw == bytes(32)
    for a in (P, P + 1, P + 5, 2**256 - 1):           # weak representations in the base field
        out = ctypes.create_string_buffer(32)
        hs.hs_modinv30(0, b(a), out)
        assert int.from_bytes(out.raw, "little") == (pow(a % P, -1, P) if a % P else 0)


def msm(hs, es, ps, c):
    out = ctypes.create_string_buffer(64)
    hs.hs_msm(b"".join(b(e) for e in es), b"".join(pb(p) for p in ps), len(es), c, out)
    return dp(out.raw)


@pytest.mark.parametrize("c", [1, 3, 4, 7, 10, 13, 16])
def test_bucket_msm_bodies(hs, c):
    rng = random.Random(30 + c)
    base = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(10)]
    for n in [0, 1, 2, 9, 20]:
        ps = [rng.choice(base + [None]) for _ in range(n)]
        es = [rng.choice([0, 1, 2, R - 1, R - 2, 2**255, rng.randrange(R), rng.randrange(R), rng.randrange(2**64)])
              for _ in range(n)]
        assert msm(hs, es, ps, c) == o.sum_exps(es, ps)
    p = base[0]
    assert msm(hs, [5, R - 5], [p, p], c) is None
    assert msm(hs, [7, 7, 7], [p, p, p], c) == o.point_mul(p, 21)
    if c == 7:
        for case in golden()["sumExps"]:
            assert msm(hs, [ux(s) for s in case["scalars"]], [up(q) for q in case["points"]], c) == up(case["result"])


def test_paired_sha256_and_decimal_writer(hs):
    """transcript.h: two messages hashed in lock-step (SHA-NI where the CPU has it) equal hashlib for every
    combination of lengths around the block and padding boundaries, with and without absorbed prefixes;
    the reciprocal-division decimal writer equals str()."""
    import ctypes, hashlib, os, random
    rng = random.Random(5)
    lens = [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 200, 447, 448, 449, 640, 1000]
    hs.hs_sha256_pair.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p]
    for la in lens:
        for lb in lens[::2] + [la]:
            a = bytes(rng.randrange(256) for _ in range(la))
            b = bytes(rng.randrange(256) for _ in range(lb))
            ba = ctypes.create_string_buffer(a + bytes(80), la + 80)
            bb = ctypes.create_string_buffer(b + bytes(80), lb + 80)
            oa, ob = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
            hs.hs_sha256_pair(ba, la, rng.randrange(3), bb, lb, rng.randrange(5), oa, ob)
            assert oa.raw == hashlib.sha256(a).digest() and ob.raw == hashlib.sha256(b).digest(), (la, lb)
    hs.hs_u256_to_dec.restype = ctypes.c_size_t
    vals = [0, 1, 9, 10, 10**19 - 1, 10**19, 10**19 + 1, 10**38 - 1, 10**38, 10**57, 10**76, 2**64 - 1, 2**64, 2**128 - 1,
            2**192, 2**256 - 1, 10**77, 10**77 - 1] + [rng.randrange(2**rng.randrange(1, 257)) for _ in range(3000)]
    out = ctypes.create_string_buffer(100)
    for v in vals:
        n = hs.hs_u256_to_dec(v.to_bytes(32, "little"), out)
        assert out.raw[:n] == str(v).encode(), v


# ---- frvec.h: the element bodies of the scalar-field vector kernels ----------------------------
def _vec(vals):
    return b"".join(b(v) for v in vals)


def _unvec(raw, n):
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)]


def test_power_vector_incl_zero_quirk(hs):
    """powerVector (Utils.hs:50-52): a^i, and all zeros for a = 0 (0^0 = 0 there)."""
    rng = random.Random(801)
    for a in (0, 1, 2, R - 1, rng.randrange(R)):
        out = ctypes.create_string_buffer(32 * 70)
        hs.hs_power_vector(b(a), 70, out)
        assert _unvec(out.raw, 70) == o.power_vector(a, 70)


@pytest.mark.parametrize("n,m", [(1, 1), (4, 2), (8, 4), (64, 1), (16, 8)])
def test_rp_polys_match_compute_lr_polys(hs, n, m):
    """computeLRPolys (MultiRangeProof/Prover.hs:133-154) element by element."""
    rng = random.Random(810 + n + m)
    vs = [rng.randrange(2**n) for _ in range(m)]
    a_l = o.reversed_encode_bit_multi(n, vs)
    a_r = o.complementary_vector(a_l)
    nm = n * m
    s_l = [rng.randrange(R) for _ in range(nm)]
    s_r = [rng.randrange(R) for _ in range(nm)]
    y, z = rng.randrange(1, R), rng.randrange(1, R)
    l0, l1, r0, r1 = o.compute_lr_polys(n, m, a_l, a_r, s_l, s_r, y, z)
    outs = [ctypes.create_string_buffer(32 * nm) for _ in range(3)]
    hs.hs_rp_polys(n, nm, bytes(a_l), _vec(s_r), b(y), b(z), *outs)
    assert _unvec(outs[0].raw, nm) == [v % R for v in l0]
    assert _unvec(outs[1].raw, nm) == [v % R for v in r0]
    assert _unvec(outs[2].raw, nm) == [v % R for v in r1]
    assert l1 == s_l


def test_ipp_round_bodies_match_oracle_trace(hs):
    """Three rounds of generateProof' (InnerProductProof/Prover.hs:52-92) at n = 8 on a random base: the folded witness
    equals the oracle's ls', rs'; the coefficient form reproduces the oracle's folded generators gs'', hs''; the
    expanded scalars of the next round give the oracle's lCommit, rCommit and commitment."""
    rng = random.Random(820)
    n = 8
    gs = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(n)]
    hsv = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(n)]
    hscale = [rng.randrange(1, R) for _ in range(n)]
    hs_scaled = [o.point_mul(p, s) for p, s in zip(hsv, hscale)]
    u = o.point_mul(o.GEN, rng.randrange(1, R))
    ls = [rng.randrange(R) for _ in range(n)]
    rs = [rng.randrange(R) for _ in range(n)]
    P = o.point_add(o.point_add(o.sum_exps(ls, gs), o.sum_exps(rs, hs_scaled)), o.point_mul(u, o.dot(ls, rs)))
    b_ls, b_rs = ctypes.create_string_buffer(_vec(ls), 32 * n), ctypes.create_string_buffer(_vec(rs), 32 * n)
    cg = ctypes.create_string_buffer(_vec([1] * n), 32 * n)
    ch = ctypes.create_string_buffer(_vec(hscale), 32 * n)
    nt = 2 * n + 1
    sc = ctypes.create_string_buffer(3 * nt * 32)
    cur_g, cur_h, cur_l, cur_r, cur_p = gs, hs_scaled, ls, rs, P
    nk, fold, x, xi = n, 0, 0, 0
    while True:
        hs.hs_ipp_round(n, nk, fold, b(x), b(xi), b_ls, b_rs, cg, ch, sc)
        if fold:
            nk //= 2
        assert _unvec(b_ls.raw, nk) == [v % R for v in cur_l] and _unvec(b_rs.raw, nk) == [v % R for v in cur_r]
        s = _unvec(sc.raw, 3 * nt)
        sums = [o.point_add(o.sum_exps(s[q * nt:q * nt + n], gs), o.sum_exps(s[q * nt + n:q * nt + 2 * n], hsv))
                for q in range(3)]
        # commitment of the folded witness under the folded base (without the bH term)
        assert sums[0] == o.point_add(o.sum_exps(cur_l, cur_g), o.sum_exps(cur_r, cur_h))
        if nk == 1:
            break
        lc, rc, xo, g2, h2, l2, r2, p2 = o.ipp_round(cur_g, cur_h, u, cur_p, cur_l, cur_r)
        h = nk // 2
        assert o.point_add(sums[1], o.point_mul(u, o.dot(cur_l[:h], cur_r[h:]))) == lc
        assert o.point_add(sums[2], o.point_mul(u, o.dot(cur_l[h:], cur_r[:h]))) == rc
        cur_g, cur_h, cur_l, cur_r, cur_p = g2, h2, l2, r2, p2
        fold, x, xi = 1, xo, pow(xo, -1, R)


@pytest.mark.parametrize("lg,k", [(0, 0), (1, 1), (3, 3), (5, 5), (3, 5)])
def test_other_exponents_closed_form(hs, lg, k):
    """mkOtherExponents (InnerProductProof/Verifier.hs:55-73): the Map's first-insert-wins walk equals the
    product over set bits, with `challenges` in reversed round order."""
    rng = random.Random(830 + lg + 7 * k)
    chal = [rng.randrange(1, R) for _ in range(k)]     # round order
    out = ctypes.create_string_buffer(32 << lg)
    hs.hs_other_exponents(lg, k, _vec(chal), out)
    assert _unvec(out.raw, 1 << lg) == o.mk_other_exponents(1 << lg, chal[::-1])
