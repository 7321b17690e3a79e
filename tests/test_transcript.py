"""CPU tests of the host transcript layer (SHA-256, decimal `show` text, digest
reduction; bulletproofs_amd/csrc/transcript.h) against the oracle's restatement
of Utils.hs:29-33,145-174 - no GPU involved."""
import hashlib
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib

R, P = o.R, o.P


@pytest.fixture(params=[0, 1])
def style(request):
    old = o.SHOW_STYLE
    o.SHOW_STYLE = "derived" if request.param == 0 else "plain"
    yield request.param
    o.SHOW_STYLE = old


def rand_point(rng):
    return o.point_mul(o.GEN, rng.randrange(1, R))


def test_oracle_raw_matches_hashlib():
    rng = random.Random(1)
    for n in [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 500, 1000, 5000]:
        raw = bytes(rng.randrange(256) for _ in range(n))
        d = int.from_bytes(hashlib.sha256(raw).digest(), "big")
        assert _lib.transcript(0, 5, raw=raw) == d % R
        assert _lib.transcript(0, 6, raw=raw) == d % P


def test_shamir_functions(style):
    rng = random.Random(2 + style)
    edge_scalars = [0, 1, 9, 10, 10**19 - 1, 10**19, 10**38, R - 1, 2**64, 2**128, 2**192]
    for it in range(40):
        A, S, T1, T2 = (rand_point(rng) for _ in range(4))
        if it == 3:
            A = None  # `O` prints as "O"
        if it == 4:
            S = (5, 10**40)  # short decimal expansions
        y = rng.choice(edge_scalars + [rng.randrange(R)])
        z = rng.choice(edge_scalars + [rng.randrange(R)])
        t = rng.choice(edge_scalars + [rng.randrange(R)])
        assert _lib.transcript(style, 0, [A, S]) == o.shamir_y(A, S)
        assert _lib.transcript(style, 1, [A, S], [y]) == o.shamir_z(A, S, y)
        assert _lib.transcript(style, 2, [A, S, T1, T2], [y, z]) == o.shamir_x(A, S, T1, T2, y, z)
        assert _lib.transcript(style, 3, [A, S, T1]) == o.shamir_x_prime(A, S, T1)
        assert _lib.transcript(style, 4, [], [y, z, t]) == o.shamir_u(y, z, t)


def test_log_base2_double_arithmetic():
    """Utils.hs:93-94 incl. Double rounding of 2^64 +/- 1."""
    for k in list(range(1, 130)) + [255, 256]:
        assert _lib.log_base2(2**k) == o.log_base2(2**k) == k
    for v in [1, 2, 3, 5, 1000, 2**64 - 1, 2**64 + 1, 2**53 + 1, R + 1, 3 * 2**70]:
        assert _lib.log_base2(v) == o.log_base2(v)


def test_portable_sha_path_matches():
    """Force the non-SHA-NI code path in a child process (BP_NO_SHANI) and compare digests."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from bulletproofs_amd import _lib; "
            "print(_lib.transcript(0, 5, raw=bytes(range(256)) * 5), _lib.transcript(1, 6, raw=b'abc'))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, BP_NO_SHANI="1")).split()
    raw = bytes(range(256)) * 5
    assert int(out[0]) == int.from_bytes(hashlib.sha256(raw).digest(), "big") % R
    assert int(out[1]) == int.from_bytes(hashlib.sha256(b"abc").digest(), "big") % P


def test_circuit_shamir_functions(style):
    """ArithmeticCircuit/Internal.hs:109-117: shamirGxGxG, shamirGs (with `O` among the points, as
    tCommits[0] = commit 0 0 always is) and the one-argument shamirZ."""
    rng = random.Random(20 + style)
    for it in range(10):
        a, b, c = (rand_point(rng) for _ in range(3))
        y = rng.randrange(R)
        assert _lib.transcript(style, 7, [a, b, c]) == o.shamir_gxgxg(a, b, c)
        assert _lib.transcript(style, 8, [], [y]) == o.ac_shamir_z(y)
        ps = [None] + [rand_point(rng) for _ in range(6)]
        if it == 1:
            ps[2] = None
        assert _lib.transcript(style, 9, ps, raw=b"\0" * len(ps)) == o.shamir_gs(ps)
    assert _lib.transcript(style, 9, [], raw=b"") == o.shamir_gs([])
