"""GPU parity tests (through the C ABI) for field arithmetic and element-wise
group operations, against the oracle and the golden vectors.  Bit-exact."""
import random

import pytest

import bp_oracle as o
from util import golden, ux, up

pytestmark = pytest.mark.gpu
P, R = o.P, o.R


@pytest.mark.parametrize("field,mod", [(0, P), (1, R), (2, P)])
def test_field_ops_exact(engine, field, mod):
    rng = random.Random(40 + field)
    edge = [0, 1, 2, mod - 1, mod - 2, 2**255 % mod, (2**32 + 977) % mod, 977, 2**128]
    a = edge * len(edge) + [rng.randrange(mod) for _ in range(2000)]
    b = [e for e in edge for _ in edge] + [rng.randrange(mod) for _ in range(2000)]
    assert engine.field_op(field, 0, a, b) == [(x + y) % mod for x, y in zip(a, b)]
    assert engine.field_op(field, 1, a, b) == [(x - y) % mod for x, y in zip(a, b)]
    assert engine.field_op(field, 2, a, b) == [(x * y) % mod for x, y in zip(a, b)]
    assert engine.field_op(field, 3, a) == [x * x % mod for x in a]
    assert engine.field_op(field, 5, a) == [(-x) % mod for x in a]
    # inversion (almost Montgomery inverse + 2^-k correction): powers of two and their negatives drive the
    # step count k to both ends of [256, 512]
    ai = a[:600] + [2**i % mod for i in range(256)] + [(mod - 2**i) % mod for i in range(0, 256, 3)] + [3, 5, (mod + 1) // 2]
    inv = engine.field_op(field, 4, ai)
    assert inv == [pow(x, -1, mod) if x else 0 for x in ai]


def test_fq_weakly_reduced_inputs(engine):
    """The kernels keep Fq values weakly reduced (any 256-bit representative); add/sub fold the wrap past
    2^256 back with 2^256 = 2^32 + 977 (mod p), twice in the rare cases these inputs force."""
    C = 2**32 + 977
    top = 2**256
    vals = [0, 1, 976, 977, 978, C - 1, C, C + 1, 2 * C - 1, 2 * C, 2 * C + 1, 2**33, 2**64 - 1, P - 1, P, P + 1,
            top - 2 * C - 1, top - 2 * C, top - C - 1, top - C, top - C + 1, top - 978, top - 977, top - 2, top - 1,
            2**255, 2**255 + C]
    rng = random.Random(77)
    vals += [rng.randrange(top) for _ in range(40)] + [top - 1 - rng.randrange(2 * C) for _ in range(20)]
    a = [x for x in vals for _ in vals]
    b = [y for _ in vals for y in vals]
    assert engine.field_op(0, 0, a, b) == [(x + y) % P for x, y in zip(a, b)]
    assert engine.field_op(0, 1, a, b) == [(x - y) % P for x, y in zip(a, b)]
    assert engine.field_op(0, 2, a, b) == [(x * y) % P for x, y in zip(a, b)]
    assert engine.field_op(0, 5, vals) == [(-x) % P for x in vals]
    assert engine.field_op(0, 4, vals) == [pow(x % P, -1, P) if x % P else 0 for x in vals]


def test_field_laws(engine):
    """test/Test/Field.hs:39-64 restated on the device for Fr and Fq."""
    rng = random.Random(41)
    for field, mod in [(1, R), (0, P)]:
        x = [rng.randrange(mod) for _ in range(256)]
        y = [rng.randrange(mod) for _ in range(256)]
        z = [rng.randrange(mod) for _ in range(256)]
        f = engine.field_op
        assert f(field, 0, x, y) == f(field, 0, y, x)
        assert f(field, 2, x, y) == f(field, 2, y, x)
        assert f(field, 0, f(field, 0, x, y), z) == f(field, 0, x, f(field, 0, y, z))
        assert f(field, 2, f(field, 2, x, y), z) == f(field, 2, x, f(field, 2, y, z))
        assert f(field, 0, x, [0] * 256) == x and f(field, 2, x, [1] * 256) == x
        assert f(field, 0, x, f(field, 5, x)) == [0] * 256
        assert f(field, 2, x, f(field, 4, x)) == [1 if v else 0 for v in x]
        assert f(field, 2, x, f(field, 0, y, z)) == f(field, 0, f(field, 2, x, y), f(field, 2, x, z))


def test_golden_field(engine):
    g = golden()
    for key, field in [("fq", 0), ("fr", 1), ("fq", 2)]:
        a = [ux(e["a"]) for e in g[key]]
        b = [ux(e["b"]) for e in g[key]]
        assert engine.field_op(field, 0, a, b) == [ux(e["add"]) for e in g[key]]
        assert engine.field_op(field, 1, a, b) == [ux(e["sub"]) for e in g[key]]
        assert engine.field_op(field, 2, a, b) == [ux(e["mul"]) for e in g[key]]
        assert engine.field_op(field, 4, a) == [ux(e["inv"]) for e in g[key]]


def test_point_mul_add_neg(engine):
    rng = random.Random(42)
    G = o.GEN
    pts = [None, G, o.point_neg(G)] + [o.point_mul(G, rng.randrange(1, R)) for _ in range(13)]
    ks = [0, 1, 2, R - 1, R - 2, R, R + 5, 2**256 - 1] + [rng.randrange(R) for _ in range(8)]
    ps = [pts[i % len(pts)] for i in range(64)]
    ss = [ks[(i * 7) % len(ks)] for i in range(64)]
    assert engine.mul(ps, ss) == [o.point_mul(p, k) for p, k in zip(ps, ss)]
    a = [p for p in pts for _ in pts]
    b = [q for _ in pts for q in pts]
    assert engine.add(a, b) == [o.point_add(p, q) for p, q in zip(a, b)]
    assert engine.add(pts, [o.point_neg(p) for p in pts]) == [None] * len(pts)
    assert engine.inv(pts) == [o.point_neg(p) for p in pts]
    assert engine.on_curve(pts + [(5, 7)]) == [True] * len(pts) + [False]
    for e in golden()["kG"]:
        assert engine.mul([G], [ux(e["k"])]) == [up(e["P"])]


def test_prop_add_sub_mod(engine):
    """test/Test/Field.hs:15-33: G*(x+y) = G*x + G*y, G*(x-y) = G*x + inv(G*y)."""
    rng = random.Random(43)
    n = 64
    x = [rng.randrange(R) for _ in range(n)]
    y = [rng.randrange(R) for _ in range(n)]
    G = [o.GEN] * n
    gx, gy = engine.mul(G, x), engine.mul(G, y)
    assert engine.mul(G, [(a + b) % R for a, b in zip(x, y)]) == engine.add(gx, gy)
    assert engine.mul(G, [(a - b) % R for a, b in zip(x, y)]) == engine.add(gx, engine.inv(gy))


def test_add_two_mul_p(engine):
    """Utils.hs:72-73, incl. commit = addTwoMulP x gen r h (Utils.hs:84-85)."""
    rng = random.Random(44)
    n = 48
    p0 = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(n)]
    p1 = [o.point_mul(o.GEN, rng.randrange(1, R)) for _ in range(n)]
    e0 = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(n)]
    e1 = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(n)]
    p1[3] = p0[3]
    p1[4] = o.point_neg(p0[4]); e1[4] = e0[4]
    p0[5] = None
    assert engine.add_two_mul_p(e0, p0, e1, p1) == [o.add_two_mul_p(a, p, c, q) for a, p, c, q in zip(e0, p0, e1, p1)]
    xs = [rng.randrange(R) for _ in range(8)]
    rs = [rng.randrange(R) for _ in range(8)]
    assert engine.add_two_mul_p(xs, [o.GEN] * 8, rs, [o.h()] * 8) == [o.commit(a, c) for a, c in zip(xs, rs)]


def test_device_decimal_text(engine):
    """The kernels' `show` digits (dec.h, k_dec_numbers: what the host transcript concatenates) equal str()
    for edge values, every power of ten and random integers, across several workgroups."""
    rng = random.Random(2024)
    vals = [0, 1, 9, 10, 2**32 - 1, 2**32, 2**64, 10**77 - 1, 10**77, o.P - 1, o.P, o.R, 2**256 - 1]
    vals += [10**e for e in range(78)] + [10**e - 1 for e in range(1, 78)]
    vals += [rng.randrange(2**rng.randrange(1, 257)) for _ in range(3000)]
    assert engine.show_numbers(vals) == [str(v) for v in vals]
    assert engine.show_numbers([]) == []
