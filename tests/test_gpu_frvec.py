"""GPU parity tests of the scalar-field vector kernels (frvec.hip) against the oracle:
powerVector / hadamard / dot (Utils.hs:50-69), computeLRPolys / computeTPoly (MultiRangeProof/Prover.hs:133-169),
and the witness side of the inner-product recursion (InnerProductProof/Prover.hs:60-61,84-85) against o.ipp_round
traces, including the workgroup reductions that the CPU-side element tests (tests/test_hostsim.py) cannot see."""
import random

import pytest

import bp_oracle as o

pytestmark = pytest.mark.gpu
R = o.R


def test_power_vector_hadamard_dot(engine):
    rng = random.Random(901)
    for a in (0, 1, 2, R - 1, rng.randrange(R)):
        for n in (1, 5, 64, 1025):
            assert engine.fr_vector_op(0, (a, n)) == o.power_vector(a, n)
    for n in (1, 63, 64, 65, 1024, 4097):
        a = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(n)]
        b = [rng.randrange(R) for _ in range(n)]
        assert engine.fr_vector_op(1, a, b) == o.hadamard(a, b)
        assert engine.fr_vector_op(2, a, b) == o.dot(a, b)


@pytest.mark.parametrize("n,m", [(1, 1), (8, 4), (64, 1), (64, 16), (32, 5)])
def test_compute_lr_polys_and_t_poly(engine, n, m):
    rng = random.Random(910 + n + m)
    vs = [rng.randrange(2**n) for _ in range(m)]
    a_l = o.reversed_encode_bit_multi(n, vs)
    a_r = o.complementary_vector(a_l)
    nm = n * m
    s_l = [rng.randrange(R) for _ in range(nm)]
    s_r = [rng.randrange(R) for _ in range(nm)]
    y, z, x = (rng.randrange(1, R) for _ in range(3))
    l0, l1, r0, r1 = o.compute_lr_polys(n, m, a_l, a_r, s_l, s_r, y, z)
    t0, t1, t2 = o.compute_t_poly(l0, l1, r0, r1)
    g_l0, g_r0, g_r1, g_ls, g_rs, g_t = engine.rp_polys(n, m, a_l, s_l, s_r, y, z, x)
    assert g_l0 == [v % R for v in l0] and g_r0 == [v % R for v in r0] and g_r1 == [v % R for v in r1]
    assert g_t == (t0 % R, t1 % R, t2 % R)
    assert g_ls == o.vadd(l0, [x * v % R for v in l1])      # MultiRangeProof/Prover.hs:90
    assert g_rs == o.vadd(r0, [x * v % R for v in r1])      # :91


def test_compute_lr_polys_y_zero_quirk(engine):
    """powerVector 0 nm is all zeros (Utils.hs:50-52), so y = 0 zeroes r1 and the y-part of r0."""
    rng = random.Random(915)
    n, m = 8, 2
    a_l = o.reversed_encode_bit_multi(n, [5, 200])
    a_r = o.complementary_vector(a_l)
    s_l = [rng.randrange(R) for _ in range(16)]
    s_r = [rng.randrange(R) for _ in range(16)]
    z, x = rng.randrange(1, R), rng.randrange(1, R)
    l0, l1, r0, r1 = o.compute_lr_polys(n, m, a_l, a_r, s_l, s_r, 0, z)
    g = engine.rp_polys(n, m, a_l, s_l, s_r, 0, z, x)
    assert g[1] == [v % R for v in r0] and g[2] == [0] * 16 == [v % R for v in r1]


@pytest.mark.parametrize("n", [2, 8, 64, 2048])
def test_ipp_witness_rounds_vs_oracle_trace(engine, n):
    """ls', rs', cL, cR round by round against the oracle's recursion on the witness, and the expanded scalars of the
    three sums: sum over generators i of a block position j must give back ls_j * (fold coefficient)."""
    rng = random.Random(920 + n)
    ls = [rng.randrange(R) for _ in range(n)]
    rs = [rng.randrange(R) for _ in range(n)]
    hscale = [rng.randrange(1, R) for _ in range(n)]
    ub = rng.randrange(1, R)
    k = n.bit_length() - 1
    xs = [rng.randrange(1, R) for _ in range(k)]
    cg, ch = [1] * n, list(hscale)
    cur_l, cur_r, nk = ls, rs, n
    for r in range(k + 1):
        g_l, g_r, dot, ucl, ucr, sc = engine.ipp_fold(ls, rs, hscale, ub, xs[:r])
        assert g_l == cur_l and g_r == cur_r
        h = nk // 2
        assert dot == o.dot(cur_l, cur_r)
        if nk >= 2:
            assert ucl == ub * o.dot(cur_l[:h], cur_r[h:]) % R      # cL = dot lsLeft rsRight  (Prover.hs:60)
            assert ucr == ub * o.dot(cur_l[h:], cur_r[:h]) % R      # cR = dot lsRight rsLeft  (:61)
        nt = 2 * n + 1
        assert sc[0][2 * n] == ub * dot % R
        for i in range(n):
            j = i & (nk - 1)
            assert sc[0][i] == cur_l[j] * cg[i] % R and sc[0][n + i] == cur_r[j] * ch[i] % R
            if nk >= 2:
                if j >= h:
                    exp = (cur_l[j - h] * cg[i] % R, 0, 0, cur_r[j - h] * ch[i] % R)
                else:
                    exp = (0, cur_r[j + h] * ch[i] % R, cur_l[j + h] * cg[i] % R, 0)
                assert (sc[1][i], sc[1][n + i], sc[2][i], sc[2][n + i]) == exp
        if r == k:
            break
        x = xs[r]
        xi = pow(x, -1, R)
        # ls' = x lsLeft + xInv lsRight, rs' = xInv rsLeft + x rsRight   (Prover.hs:84-85)
        cur_l = o.vadd([x * v % R for v in cur_l[:h]], [xi * v % R for v in cur_l[h:]])
        cur_r = o.vadd([xi * v % R for v in cur_r[:h]], [x * v % R for v in cur_r[h:]])
        # gs'' = xInv gsLeft + x gsRight, hs'' = x hsLeft + xInv hsRight   (:81-82) as coefficients
        for i in range(n):
            lo = (i & (nk - 1)) < h
            cg[i] = cg[i] * (xi if lo else x) % R
            ch[i] = ch[i] * (x if lo else xi) % R
        nk = h
