"""CPU oracle for the elliptic-curve hot path of sdiehl/bulletproofs v1.1.0.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (bulletproofs_amd/,
the C-ABI library) may import, link or execute this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
the checker.

This is a plain big-integer restatement of the reference's algorithm.  Every
function cites the reference file:line it follows (paths relative to
/root/reference).

PARITY STATUS
-------------
* Numeric EC results (field ops, point add/double/mul, sumExps, commit, IPP
  folds): the reference delegates them to un-vendored third-party packages
  (elliptic-curve-0.3.0, galois-field-1.0.0; pins stack.yaml:3-4).  The
  group law of secp256k1 is standard and has a unique affine answer; this
  file is pinned against OpenSSL libcrypto's secp256k1 (tests/test_oracle.py)
  and published k*G constants.
* Fiat-Shamir transcripts hash the Haskell `show` text of points/scalars
  (Utils.hs:32-33,145-174).  That text format, the square-root branch picked
  by `sr` (Utils.hs:47) and the RNG stream are defined outside the reference
  tree and the reference holds no golden vector (all tests are QuickCheck
  properties): for these three details parity is UNPINNED.  Each is isolated
  behind one switch (SHOW_STYLE, SQRT_BRANCH; blinders are explicit inputs).
"""
import hashlib
import math

# --------------------------------------------------------------------------
# secp256k1 (a-1: Data.Curve.Weierstrass.SECP256K1, imported at Utils.hs:9)
# --------------------------------------------------------------------------
P = 2**256 - 2**32 - 977                      # base field Fq
R = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141  # _r
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
O = None                                      # `O` / mempty
GEN = (GX, GY)                                # `gen`

# U1: text produced by `show` on PA / Fr.
#   "derived": show (A x y) = "A (P <x>) (P <y>)", show O = "O", show s = "P <s>"
#   "plain"  : show (A x y) = "A <x> <y>",         show O = "O", show s = "<s>"
SHOW_STYLE = "derived"
# U2: branch of the square root returned by `sr` (Utils.hs:47).
#   "p34": y = (x^3+7)^((p+1)/4);  "neg": p - that.
SQRT_BRANCH = "p34"


def is_on_curve(pt):
    """`def` (Utils.hs:42)."""
    if pt is O:
        return True
    x, y = pt
    return (y * y - x * x * x - 7) % P == 0


def point_neg(pt):
    """`inv` (RangeProof/Internal.hs:187)."""
    if pt is O:
        return O
    return (pt[0], (-pt[1]) % P)


def point_add(p1, p2):
    """`<>` / `add` on affine points: total group law."""
    if p1 is O:
        return p2
    if p2 is O:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return O
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def _jac_double(X, Y, Z):
    if Y == 0 or Z == 0:
        return (0, 1, 0)
    S = 4 * X * Y * Y % P
    M = 3 * X * X % P
    X3 = (M * M - 2 * S) % P
    Y3 = (M * (S - X3) - 8 * pow(Y, 4, P)) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    r = (S2 - Y1) % P
    if H == 0:
        if r == 0:
            return _jac_double(X1, Y1, Z1)
        return (0, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def point_mul(pt, k):
    """`mul pt k` (double-and-add scalar multiplication, Utils.hs:73).

    Evaluated in Jacobian coordinates and normalised once; the affine
    result is the same group element the reference's affine ladder yields.
    """
    k %= R
    if pt is O or k == 0:
        return O
    x, y = pt
    X, Y, Z = 0, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_double(X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(X, Y, Z, x, y)
    if Z == 0:
        return O
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


# --------------------------------------------------------------------------
# Transcript text and the random oracle (a-18)
# --------------------------------------------------------------------------
def show_fr(s):
    """`show (s :: Fr)` (Utils.hs:153,165,174)."""
    if SHOW_STYLE == "derived":
        return b"P %d" % s
    return b"%d" % s


def show_point(pt):
    """`pointToBS = show` (Utils.hs:32-33)."""
    if pt is O:
        return b"O"
    if SHOW_STYLE == "derived":
        return b"A (P %d) (P %d)" % pt
    return b"A %d %d" % pt


SHOW_R = b"%d" % R                            # `show _r`


def oracle(bs, modulus=R):
    """`oracle = fromInteger . integerDigest . sha256` (Utils.hs:29-30).

    Type-directed in the reference: into Fr everywhere except generateH,
    which lands in Fq (Utils.hs:46)."""
    return int.from_bytes(hashlib.sha256(bs).digest(), "big") % modulus


def fq_sqrt(a):
    """`sr` (Utils.hs:47); Nothing -> None."""
    a %= P
    if a == 0:
        return 0
    y = pow(a, (P + 1) // 4, P)
    if y * y % P != a:
        return None
    return y if SQRT_BRANCH == "p34" else P - y


def generate_h(extra=b""):
    """`generateH` (Utils.hs:38-47)."""
    while True:
        x = oracle(show_point(GEN) + extra, P)
        y = fq_sqrt(x**3 + 7)
        if y is not None and is_on_curve((x, y)):
            return (x, y)
        extra = b"1" + extra


_H = None
_GS = []
_HS = []


def h():
    """`h = generateH ""` (Utils.hs:14-15)."""
    global _H
    if _H is None:
        _H = generate_h()
    return _H


def gen_scalar_g(i):
    """Discrete log of gs!!(i-1) w.r.t. gen: oracle(show i <> show gen) (Utils.hs:20)."""
    return oracle(b"%d" % i + show_point(GEN))


def gen_scalar_h(i):
    """Discrete log of hs!!(i-1) w.r.t. gen (Utils.hs:25)."""
    return oracle(b"%d" % i + show_point(h()))


def gs(n):
    """`take n gs` (Utils.hs:19-20); index i starts at 1."""
    while len(_GS) < n:
        _GS.append(point_mul(GEN, gen_scalar_g(len(_GS) + 1)))
    return _GS[:n]


def hs(n):
    """`take n hs` (Utils.hs:24-25)."""
    while len(_HS) < n:
        _HS.append(point_mul(GEN, gen_scalar_h(len(_HS) + 1)))
    return _HS[:n]


def reset_generators():
    """Drop cached h/gs/hs (after changing SHOW_STYLE / SQRT_BRANCH)."""
    global _H
    _H = None
    _GS.clear()
    _HS.clear()


def shamir_y(a, s):
    """Utils.hs:145-148."""
    return oracle(SHOW_R + show_point(a) + show_point(s))


def shamir_z(a, s, y):
    """Utils.hs:150-153."""
    return oracle(SHOW_R + show_point(a) + show_point(s) + show_fr(y))


def shamir_x(a, s, t1, t2, y, z):
    """Utils.hs:155-165."""
    return oracle(SHOW_R + show_point(a) + show_point(s) + show_point(t1)
                  + show_point(t2) + show_fr(y) + show_fr(z))


def shamir_x_prime(commitment_lr, l, r):
    """`shamirX'` (Utils.hs:167-170): order is L, R, then the commitment."""
    return oracle(SHOW_R + show_point(l) + show_point(r) + show_point(commitment_lr))


def shamir_u(t_blinding, mu, t):
    """Utils.hs:172-174."""
    return oracle(SHOW_R + show_fr(t_blinding) + show_fr(mu) + show_fr(t))


# --------------------------------------------------------------------------
# Vector helpers (Utils.hs:49-139)
# --------------------------------------------------------------------------
def power_vector(a, n):
    """Utils.hs:50-52, including the 0^0 = 0 special case."""
    a %= R
    out, cur = [], 1
    for i in range(n):
        out.append(0 if (i == 0 and a == 0) else cur)
        cur = cur * a % R
    return out


def hadamard(a, b):
    """Utils.hs:55-57."""
    if len(a) != len(b):
        raise ValueError("Vector sizes must match")
    return [x * y % R for x, y in zip(a, b)]


def dot(a, b):
    """Utils.hs:60-61."""
    return sum(hadamard(a, b)) % R


def vadd(a, b):
    """`^+^` (Utils.hs:64-65): zipWith truncates."""
    return [(x + y) % R for x, y in zip(a, b)]


def vsub(a, b):
    """`^-^` (Utils.hs:68-69)."""
    return [(x - y) % R for x, y in zip(a, b)]


def add_two_mul_p(e0, p0, e1, p1):
    """`addTwoMulP` (Utils.hs:72-73)."""
    return point_add(point_mul(p0, e0), point_mul(p1, e1))


def sum_exps(es, ps):
    """`sumExps` (Utils.hs:76-80): pairs from the left, right-nested sum,
    truncation to the shorter list, [] -> O."""
    n = min(len(es), len(ps))
    acc = O
    # right-nested fold evaluated from the right so that the association
    # order equals the reference's (the group is associative anyway).
    i = n - (n % 2)
    if n % 2:
        acc = point_mul(ps[n - 1], es[n - 1])
    while i > 0:
        i -= 2
        acc = point_add(add_two_mul_p(es[i], ps[i], es[i + 1], ps[i + 1]), acc)
    return acc


def commit(x, r):
    """`commit x r = addTwoMulP x gen r h` (Utils.hs:84-85)."""
    return add_two_mul_p(x, GEN, r, h())


def is_log_base2(x):
    """Utils.hs:87-91."""
    if x == 1:
        return True
    if x == 0 or x % 2 != 0:
        return False
    return is_log_base2(x // 2)


def log_base2(x):
    """`floor . logBase 2.0 . fromIntegral` (Utils.hs:93-94): Double arithmetic.
    logBase 2.0 0 is -Infinity and GHC's `floor` of it on Integer is -2^1024
    (the decoded mantissa shifted by the decoded exponent)."""
    if x == 0:
        return -(2**1024)
    return int(math.floor(math.log(float(x)) / math.log(2.0)))


def _as_int(x):
    """`fromIntegral :: Integer -> Int` (64-bit wrap), as `take` / `drop` /
    `replicate` counts get it (RangeProof/Internal.hs:200, Utils.hs:103)."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def log_base2_m(x):
    """Utils.hs:96-100."""
    return log_base2(x) if is_log_base2(x) else None


def vslice(n, j, vs):
    """`slice n j vs` (Utils.hs:102-103); j is 1-based; `take` / `drop` of a
    non-positive count take / drop nothing."""
    a = max(_as_int((j - 1) * n), 0)
    return vs[a: a + max(_as_int(j * n - (j - 1) * n), 0)]


def log2_ceil(x):
    """Utils.hs:124-130 (x >= 1)."""
    floor_log = x.bit_length() - 1
    ctz = (x & -x).bit_length() - 1
    return floor_log + (1 if ctz < floor_log else 0)


# --------------------------------------------------------------------------
# Range-proof helpers (RangeProof/Internal.hs)
# --------------------------------------------------------------------------
def reversed_encode_bit(n, v):
    """RangeProof/Internal.hs:73-79: n bits of v, least significant first."""
    bits = [int(c) for c in bin(v)[2:]]
    bits = [0] * (n - len(bits)) + bits           # fillWithZeros (:93-96)
    return bits[::-1]


def reversed_encode_bit_multi(n, vs):
    """RangeProof/Internal.hs:81-82."""
    out = []
    for v in vs:
        out += reversed_encode_bit(n, v)
    return out


def complementary_vector(a_l):
    """RangeProof/Internal.hs:87-88."""
    return [(v - 1) % R for v in a_l]


def commit_bit_vectors(a_blinding, s_blinding, a_l, a_r, s_l, s_r):
    """RangeProof/Internal.hs:126-149."""
    n = max(len(a_l), len(a_r), len(s_l), len(s_r))
    g, hh = gs(n), hs(n)
    a_commit = point_add(point_add(point_mul(h(), a_blinding), sum_exps(a_l, g)), sum_exps(a_r, hh))
    s_commit = point_add(point_add(point_mul(h(), s_blinding), sum_exps(s_l, g)), sum_exps(s_r, hh))
    return a_commit, s_commit


def delta(n, m, y, z):
    """RangeProof/Internal.hs:152-157."""
    nm = n * m
    first = (z - z * z) * dot(power_vector(1, nm), power_vector(y, nm))
    second = 0
    d2 = dot(power_vector(1, n), power_vector(2, n))
    for j in range(1, m + 1):
        second += pow(z, j + 2, R) * d2
    return (first - second) % R


def check_ranges(n, vs):
    """RangeProof/Internal.hs:164-165."""
    return all(0 <= v < 2**n for v in vs)


def hs_prime(y, nm):
    """`hs' = zipWith (\\yi hi -> hi `mul` recip yi) (powerVector y nm) hs`
    (MultiRangeProof/Prover.hs:107, Verifier.hs:85)."""
    # `hs` is an infinite list: the zip is as long as powerVector y nm (empty for nm <= 0)
    return [point_mul(hi, pow(yi, -1, R)) for yi, hi in zip(power_vector(y, nm), hs(max(nm, 0)))]


def compute_lr_commitment(n, m, a_commit, s_commit, t, t_blinding, mu, x, y, z, hsp):
    """RangeProof/Internal.hs:169-206."""
    nm = n * m
    gs_sum = O
    for g in gs(max(_as_int(nm), 0)):          # take (fromIntegral nm) gs   (:200)
        gs_sum = point_add(gs_sum, g)
    h_exp = [z * yi % R for yi in power_vector(y, nm)]
    u = point_mul(GEN, shamir_u(t_blinding, mu, t))
    acc = a_commit
    acc = point_add(acc, point_mul(s_commit, x))
    acc = point_add(acc, point_neg(point_mul(gs_sum, z)))
    acc = point_add(acc, sum_exps(h_exp, hsp))
    inner = O
    for j in range(1, m + 1):
        h_exp_j = [pow(z, j + 1, R) * t2 % R for t2 in power_vector(2, n)]
        inner = point_add(inner, sum_exps(h_exp_j, vslice(n, j, hsp)))
    acc = point_add(acc, inner)
    acc = point_add(acc, point_neg(point_mul(h(), mu)))
    acc = point_add(acc, point_mul(u, t))
    return acc


# --------------------------------------------------------------------------
# Inner-product argument (InnerProductProof/{Prover,Verifier}.hs)
# --------------------------------------------------------------------------
def ipp_round(b_gs, b_hs, b_h, commitment_lr, ls, rs):
    """One recursion step of generateProof' (InnerProductProof/Prover.hs:52-92).

    Returns (lCommit, rCommit, x, gs'', hs'', ls', rs', commitmentLR')."""
    n2 = len(ls) // 2
    ls_l, ls_r = ls[:n2], ls[n2:]
    rs_l, rs_r = rs[:n2], rs[n2:]
    gs_l, gs_r = b_gs[:n2], b_gs[n2:]
    hs_l, hs_r = b_hs[:n2], b_hs[n2:]
    c_l = dot(ls_l, rs_r)
    c_r = dot(ls_r, rs_l)
    l_commit = point_add(point_add(sum_exps(ls_l, gs_r), sum_exps(rs_r, hs_l)), point_mul(b_h, c_l))
    r_commit = point_add(point_add(sum_exps(ls_r, gs_l), sum_exps(rs_l, hs_r)), point_mul(b_h, c_r))
    x = shamir_x_prime(commitment_lr, l_commit, r_commit)
    x_inv = pow(x, -1, R)
    gs2 = [add_two_mul_p(x_inv, a, x, b) for a, b in zip(gs_l, gs_r)]
    hs2 = [add_two_mul_p(x, a, x_inv, b) for a, b in zip(hs_l, hs_r)]
    ls2 = vadd([x * v % R for v in ls_l], [x_inv * v % R for v in ls_r])
    rs2 = vadd([x_inv * v % R for v in rs_l], [x * v % R for v in rs_r])
    c2 = point_add(point_add(point_mul(l_commit, x * x % R), point_mul(r_commit, x_inv * x_inv % R)),
                   commitment_lr)
    return l_commit, r_commit, x, gs2, hs2, ls2, rs2, c2


def ipp_generate_proof(b_gs, b_hs, b_h, commitment_lr, ls, rs, trace=None):
    """`generateProof` (InnerProductProof/Prover.hs:17-50).

    Returns dict(lCommits, rCommits, l, r).  Vectors are truncated the way
    Haskell pattern matching/zipWith would for equal-length inputs only;
    the reference always calls it with equal power-of-two lengths."""
    l_commits, r_commits = [], []
    while True:
        if len(ls) == 0 and len(rs) == 0:
            return dict(lCommits=[], rCommits=[], l=0, r=0)
        if len(ls) == 1 and len(rs) == 1:
            return dict(lCommits=l_commits, rCommits=r_commits, l=ls[0] % R, r=rs[0] % R)
        lc, rc, x, b_gs, b_hs, ls, rs, commitment_lr = ipp_round(b_gs, b_hs, b_h, commitment_lr, ls, rs)
        if trace is not None:
            trace.append(dict(L=lc, R=rc, x=x, P=commitment_lr, ls=list(ls), rs=list(rs),
                              gs=list(b_gs), hs=list(b_hs)))
        l_commits.append(lc)
        r_commits.append(rc)


def mk_challenges(l_commits, r_commits, commitment_lr):
    """InnerProductProof/Verifier.hs:40-53.  Challenges come back in REVERSE
    round order (x:xs)."""
    xs, xs_inv, acc = [], [], commitment_lr
    for li, ri in zip(l_commits, r_commits):
        x = shamir_x_prime(acc, li, ri)
        x_inv = pow(x, -1, R)
        acc = point_add(point_add(point_mul(li, x * x % R), point_mul(ri, x_inv * x_inv % R)), acc)
        xs.insert(0, x)
        xs_inv.insert(0, x_inv)
    return xs, xs_inv, acc


def mk_other_exponents(n, challenges):
    """InnerProductProof/Verifier.hs:55-73 (Map keyed by index; first insert wins)."""
    prod = 1
    for c in challenges:
        prod = prod * c % R
    acc = {0: pow(prod, -1, R)}
    n2 = n // 2
    lg = log_base2(n) if n > 0 else 0
    for i in range(n2):
        for j in range(lg):
            i1 = 2**j + i
            if i1 >= n or i1 in acc:
                continue
            acc[i1] = acc[i] * challenges[j] * challenges[j] % R
    return [acc[k] for k in sorted(acc)]


def ipp_verify_proof(n, b_gs, b_hs, b_h, commitment_lr, proof):
    """InnerProductProof/Verifier.hs:18-38."""
    challenges, _inv, c = mk_challenges(proof["lCommits"], proof["rCommits"], commitment_lr)
    other = mk_other_exponents(n, challenges)
    gs_commit = sum_exps(other, b_gs)
    hs_commit = sum_exps(other[::-1], b_hs)
    l, r = proof["l"], proof["r"]
    c_proof = point_add(point_add(point_mul(gs_commit, l), point_mul(hs_commit, r)),
                        point_mul(b_h, l * r % R))
    return c == c_proof


# --------------------------------------------------------------------------
# Multi range proof (MultiRangeProof/{Prover,Verifier}.hs)
# --------------------------------------------------------------------------
def compute_lr_polys(n, m, a_l, a_r, s_l, s_r, y, z):
    """MultiRangeProof/Prover.hs:133-154."""
    nm = n * m
    z1nm = [z * v % R for v in power_vector(1, nm)]
    l0 = vsub(a_l, z1nm)
    l1 = [v % R for v in s_l]
    ynm = power_vector(y, nm)
    acc = [0] * nm
    for j in range(1, m + 1):
        it = [pow(z, j + 1, R) * v % R for v in
              (power_vector(0, (j - 1) * n) + power_vector(2, n) + power_vector(0, (m - j) * n))]
        acc = vadd(it, acc)
    r0 = vadd(hadamard(ynm, vadd(a_r, z1nm)), acc)
    r1 = hadamard(ynm, s_r)
    return l0, l1, r0, r1


def compute_t_poly(l0, l1, r0, r1):
    """MultiRangeProof/Prover.hs:160-169."""
    t0 = dot(l0, r0)
    t2 = dot(l1, r1)
    t1 = (dot(vadd(l0, l1), vadd(r0, r1)) - t0 - t2) % R
    return t0, t1, t2


class RangeProofError(Exception):
    """RangeProof/Internal.hs:40-45."""


def mrp_generate_proof_unsafe(upper_bound, vs_and_blindings, s_l, s_r, a_blinding, s_blinding,
                              t1_blinding, t2_blinding, trace=None):
    """`generateProofUnsafe` (MultiRangeProof/Prover.hs:47-123).

    U3: the reference draws sL, sR (chooseBlindingVectors, Utils.hs:135-139),
    then aBlinding, sBlinding (:69-70), then t1Blinding, t2Blinding (:81-82)
    from MonadRandom; here they are explicit arguments in that draw order."""
    n = log_base2(upper_bound)
    m = len(vs_and_blindings)
    nm = n * m
    vs = [v % R for v, _ in vs_and_blindings]
    v_blindings = [b % R for _, b in vs_and_blindings]
    a_l = reversed_encode_bit_multi(n, vs)
    a_r = complementary_vector(a_l)
    s_l = [v % R for v in s_l]
    s_r = [v % R for v in s_r]
    a_commit, s_commit = commit_bit_vectors(a_blinding, s_blinding, a_l, a_r, s_l, s_r)
    y = shamir_y(a_commit, s_commit)
    z = shamir_z(a_commit, s_commit, y)
    l0, l1, r0, r1 = compute_lr_polys(n, m, a_l, a_r, s_l, s_r, y, z)
    t0, t1, t2 = compute_t_poly(l0, l1, r0, r1)
    t1_commit = commit(t1, t1_blinding)
    t2_commit = commit(t2, t2_blinding)
    x = shamir_x(a_commit, s_commit, t1_commit, t2_commit, y, z)
    ls = vadd(l0, [x * v % R for v in l1])
    rs = vadd(r0, [x * v % R for v in r1])
    t = (t0 + t1 * x + t2 * x * x) % R
    if t != dot(ls, rs):
        raise AssertionError("Error on: t = dot l r")
    t_blinding = (sum(pow(z, j + 1, R) * vb for vb, j in zip(v_blindings, range(1, m + 1)))
                  + t2_blinding * x * x + t1_blinding * x) % R
    mu = (a_blinding + s_blinding * x) % R
    u = point_mul(GEN, shamir_u(t_blinding, mu, t))
    hsp = hs_prime(y, nm)
    commitment_lr = compute_lr_commitment(n, m, a_commit, s_commit, t, t_blinding, mu, x, y, z, hsp)
    if trace is not None:
        trace.update(y=y, z=z, x=x, u=u, P=commitment_lr, ls=list(ls), rs=list(rs), rounds=[])
    product_proof = ipp_generate_proof(gs(nm), hsp, u, commitment_lr, ls, rs,
                                       trace=None if trace is None else trace["rounds"])
    return dict(tBlinding=t_blinding, mu=mu, t=t, aCommit=a_commit, sCommit=s_commit,
                t1Commit=t1_commit, t2Commit=t2_commit, productProof=product_proof)


def mrp_generate_proof(upper_bound, vs_and_blindings, s_l, s_r, a_blinding, s_blinding,
                       t1_blinding, t2_blinding):
    """`generateProof` (MultiRangeProof/Prover.hs:19-43): validation + padding
    with (0,0) to the next power of two."""
    if not upper_bound < R:
        raise RangeProofError(("UpperBoundTooLarge", upper_bound))
    n = log_base2_m(upper_bound)
    if n is None or log_base2_m(n) is None:
        raise RangeProofError(("NNotPowerOf2", upper_bound))
    vs = [v for v, _ in vs_and_blindings]
    if not check_ranges(n, vs):
        raise RangeProofError(("ValuesNotInRange", vs))
    m = len(vs_and_blindings)
    padded = list(vs_and_blindings) + [(0, 0)] * (2**log2_ceil(m) - m)
    return mrp_generate_proof_unsafe(upper_bound, padded, s_l, s_r, a_blinding, s_blinding,
                                     t1_blinding, t2_blinding)


def verify_t_poly(n, v_commits, proof, x, y, z):
    """MultiRangeProof/Verifier.hs:44-65."""
    m = len(v_commits)
    lhs = commit(proof["t"], proof["tBlinding"])
    rhs = sum_exps([z * z * v % R for v in power_vector(z, m)], v_commits)
    rhs = point_add(rhs, point_mul(GEN, delta(n, m, y, z)))
    rhs = point_add(rhs, point_mul(proof["t1Commit"], x))
    rhs = point_add(rhs, point_mul(proof["t2Commit"], x * x % R))
    return lhs == rhs


def verify_lr_commitment(n, m, proof, x, y, z):
    """MultiRangeProof/Verifier.hs:68-88."""
    nm = n * m
    hsp = hs_prime(y, nm)
    u = point_mul(GEN, shamir_u(proof["tBlinding"], proof["mu"], proof["t"]))
    commitment_lr = compute_lr_commitment(n, m, proof["aCommit"], proof["sCommit"], proof["t"],
                                          proof["tBlinding"], proof["mu"], x, y, z, hsp)
    # bGs = gs is an infinite list (Verifier.hs:80); mkOtherExponents yields at most
    # max(nm, 1) exponents (keys < nm, and key 0 always), so that prefix is all `sumExps` reads
    return ipp_verify_proof(nm, gs(max(nm, 1)), hsp, u, commitment_lr, proof["productProof"])


def mrp_verify_proof(upper_bound, v_commits, proof):
    """`verifyProof` (MultiRangeProof/Verifier.hs:19-39)."""
    y = shamir_y(proof["aCommit"], proof["sCommit"])
    z = shamir_z(proof["aCommit"], proof["sCommit"], y)
    x = shamir_x(proof["aCommit"], proof["sCommit"], proof["t1Commit"], proof["t2Commit"], y, z)
    n = log_base2(upper_bound)
    m = len(v_commits)
    if m == 0:
        # log2Ceil 0 = -1 (Utils.hs:124-130) and `2 ^ (-1)` raises in GHC
        raise ArithmeticError("Negative exponent")
    padded = list(v_commits) + [O] * (2**log2_ceil(m) - m)
    m2 = len(padded)
    return verify_t_poly(n, padded, proof, x, y, z) and verify_lr_commitment(n, m2, proof, x, y, z)


def rp_generate_proof(upper_bound, v, v_blinding, *blinders):
    """RangeProof/Prover.hs:15-22 (m = 1 wrapper)."""
    return mrp_generate_proof(upper_bound, [(v, v_blinding)], *blinders)


def rp_verify_proof(upper_bound, v_commit, proof):
    """RangeProof/Verifier.hs:17-24."""
    return mrp_verify_proof(upper_bound, [v_commit], proof)


# --------------------------------------------------------------------------
# Blinder drawing helper for tests (U3): a documented deterministic stand-in
# for MonadRandom, NOT the GHC stream.
# --------------------------------------------------------------------------
def draw_blinders(rng, nm):
    """Draw (sL, sR, aBlinding, sBlinding, t1Blinding, t2Blinding) with the
    ranges of Utils.hs:135-139 and MultiRangeProof/Prover.hs:67; `rng` is a
    random.Random."""
    s_l = [rng.randint(1, 2**nm - 1) % R for _ in range(nm)]
    s_r = [rng.randint(1, 2**nm - 1) % R for _ in range(nm)]
    rest = [rng.randint(1, R - 1) for _ in range(4)]
    return (s_l, s_r, *rest)


# --------------------------------------------------------------------------
# Arithmetic circuits (ArithmeticCircuit/{Internal,Prover,Verifier}.hs) -- a-17
# --------------------------------------------------------------------------
def pad_to_nearest_power_of_two(xs):
    """Utils.hs:107-121: [] stays []; otherwise zero-extend to 2^log2Ceil(length)."""
    if not xs:
        return []
    return list(xs) + [0] * (2**log2_ceil(len(xs)) - len(xs))


def pad_circuit(circuit):
    """ArithmeticCircuit/Internal.hs:76-87.  circuit = dict(wL, wR, wO, commitmentWeights, cs)."""
    return dict(circuit, wL=[pad_to_nearest_power_of_two(r) for r in circuit["wL"]],
                wR=[pad_to_nearest_power_of_two(r) for r in circuit["wR"]],
                wO=[pad_to_nearest_power_of_two(r) for r in circuit["wO"]])


def pad_assignment(asg):
    """ArithmeticCircuit/Internal.hs:91-97.  asg = dict(aL, aR, aO)."""
    return dict(aL=pad_to_nearest_power_of_two(asg["aL"]), aR=pad_to_nearest_power_of_two(asg["aR"]),
                aO=pad_to_nearest_power_of_two(asg["aO"]))


def ac_delta(n, y, zw_l, zw_r):
    """ArithmeticCircuit/Internal.hs:99-100."""
    return dot(hadamard(power_vector(pow(y, -1, R), n), zw_r), zw_l)


def commit_bit_vector(v_blinding, v_l, v_r):
    """ArithmeticCircuit/Internal.hs:102-107 (vLG <> vRH <> vBlindingH)."""
    n = max(len(v_l), len(v_r))
    return point_add(point_add(sum_exps(v_l, gs(n)), sum_exps(v_r, hs(n))), point_mul(h(), v_blinding))


def shamir_gxgxg(p1, p2, p3):
    """ArithmeticCircuit/Internal.hs:109-111."""
    return oracle(SHOW_R + show_point(p1) + show_point(p2) + show_point(p3))


def shamir_gs(ps):
    """ArithmeticCircuit/Internal.hs:113-114."""
    return oracle(SHOW_R + b"".join(show_point(p) for p in ps))


def ac_shamir_z(z):
    """ArithmeticCircuit/Internal.hs:116-117 (the prover/verifier hide Utils.shamirZ)."""
    return oracle(SHOW_R + show_fr(z))


def evaluate_polynomial(n, poly, x):
    """ArithmeticCircuit/Internal.hs:123-128."""
    acc = [0] * n
    for idx, e in enumerate(poly):
        acc = vadd(acc, [pow(x, idx, R) * v % R for v in e])
    return acc


def multiply_poly(l, r):
    """ArithmeticCircuit/Internal.hs:130-139: coefficients by ascending degree i + j."""
    acc = {}
    for i, li in enumerate(l):
        for j, rj in enumerate(r):
            acc[i + j] = (acc.get(i + j, 0) + dot(li, rj)) % R
    return [acc[k] for k in sorted(acc)]


def vector_matrix_product(v, m):
    """ArithmeticCircuit/Internal.hs:146-148: v * M, one entry per column (transpose truncates to
    the shortest row only when rows are ragged; the circuits here are rectangular)."""
    cols = max((len(row) for row in m), default=0)
    out = []
    for c in range(cols):
        col = [row[c] for row in m if c < len(row)]
        out.append(sum(a * b for a, b in zip(v, col)) % R)
    return out


def vector_matrix_product_t(v, m):
    """ArithmeticCircuit/Internal.hs:150-152: one entry per row."""
    return [sum(a * b for a, b in zip(v, row)) % R for row in m]


def insert_at(z, y, xs):
    """ArithmeticCircuit/Internal.hs:165-168."""
    return list(xs[:z]) + [y] + list(xs[z:])


def ac_lr_commitment(n, proof_pts, x, y, zw_l, zw_r, zw_o, mu, t, u, hsp):
    """`commitmentLR` (ArithmeticCircuit/Prover.hs:77-85 == Verifier.hs:63-71)."""
    ai, ao, s = proof_pts
    ys = power_vector(y, n)
    g_exp = [x * v % R for v in hadamard(power_vector(pow(y, -1, R), n), zw_r)]
    h_exp = vsub(vadd([x * v % R for v in zw_l], zw_o), ys)
    acc = point_mul(ai, x)
    acc = point_add(acc, point_mul(ao, x * x % R))
    acc = point_add(acc, point_mul(s, pow(x, 3, R)))
    acc = point_add(acc, sum_exps(g_exp, gs(n)))
    acc = point_add(acc, sum_exps(h_exp, hsp))
    acc = point_add(acc, point_neg(point_mul(h(), mu)))
    acc = point_add(acc, point_mul(u, t))
    return acc


def ac_generate_proof(circuit, witness, ai_blinding, ao_blinding, s_blinding, s_l, s_r, t_randoms, trace=None):
    """`generateProof` (ArithmeticCircuit/Prover.hs:16-117).

    witness = dict(assignment=dict(aL, aR, aO), commitments, commitBlinders).
    U3 draw order of the reference: aiBlinding, aoBlinding, sBlinding (:25-27), sL, sR (:32),
    then five tBlindings (:57) which land at indices 1, 3, 4, 5, 6 (0 at indices 0 and 2)."""
    c = pad_circuit(circuit)
    w_l, w_r, w_o = c["wL"], c["wR"], c["wO"]
    asg = pad_assignment(witness["assignment"])
    a_l, a_r, a_o = ([v % R for v in asg[k]] for k in ("aL", "aR", "aO"))
    q_len = len(c["commitmentWeights"])
    n = len(a_l)
    ai_commit = commit_bit_vector(ai_blinding, a_l, a_r)
    ao_commit = commit_bit_vector(ao_blinding, a_o, [])
    s_l = [v % R for v in s_l]
    s_r = [v % R for v in s_r]
    s_commit = commit_bit_vector(s_blinding, s_l, s_r)
    y = shamir_gxgxg(ai_commit, ao_commit, s_commit)
    z = ac_shamir_z(y)
    ys = power_vector(y, n)
    zs = power_vector(z, q_len + 1)[1:]
    zw_l = vector_matrix_product(zs, w_l)
    zw_r = vector_matrix_product(zs, w_r)
    zw_o = vector_matrix_product(zs, w_o)
    y_inv_n = power_vector(pow(y, -1, R), n)
    # computePolynomials (:104-117)
    l_poly = [[0] * n, vadd(a_l, hadamard(y_inv_n, zw_r)), a_o, s_l]
    r_poly = [vsub(zw_o, ys), vadd(hadamard(ys, a_r), zw_l), [0] * n, hadamard(ys, s_r)]
    t_poly = multiply_poly(l_poly, r_poly)
    t_blindings = insert_at(2, 0, [0] + [v % R for v in t_randoms])
    t_commits = [commit(tc, tb) for tc, tb in zip(t_poly, t_blindings)]
    x = shamir_gs(t_commits)
    ls = evaluate_polynomial(n, l_poly, x)
    rs = evaluate_polynomial(n, r_poly, x)
    t = dot(ls, rs)
    commit_times_weights = vector_matrix_product_t(witness["commitBlinders"], c["commitmentWeights"])
    z_gamma = dot(zs, commit_times_weights)
    t_blinding = (sum(b * pow(x, i, R) for i, b in enumerate(t_blindings)) + x * x * z_gamma) % R
    mu = (ai_blinding * x + ao_blinding * x * x + s_blinding * pow(x, 3, R)) % R
    u = point_mul(GEN, shamir_u(t_blinding, mu, t))
    hsp = [point_mul(hi, yi) for hi, yi in zip(hs(n), y_inv_n)]
    commitment_lr = ac_lr_commitment(n, (ai_commit, ao_commit, s_commit), x, y, zw_l, zw_r, zw_o, mu, t, u, hsp)
    if trace is not None:
        trace.update(y=y, z=z, x=x, tPoly=t_poly, ls=list(ls), rs=list(rs), P=commitment_lr, u=u)
    product_proof = ipp_generate_proof(gs(n), hsp, u, commitment_lr, ls, rs)
    return dict(tBlinding=t_blinding, mu=mu, t=t, aiCommit=ai_commit, aoCommit=ao_commit, sCommit=s_commit,
                tCommits=t_commits, productProof=product_proof)


def ac_verify_proof(v_commits, proof, circuit):
    """`verifyProof` (ArithmeticCircuit/Verifier.hs:16-71)."""
    c = pad_circuit(circuit)
    w_l, w_r, w_o = c["wL"], c["wR"], c["wO"]
    n = len(w_l[0])
    q_len = len(w_l)
    x = shamir_gs(proof["tCommits"])
    y = shamir_gxgxg(proof["aiCommit"], proof["aoCommit"], proof["sCommit"])
    z = ac_shamir_z(y)
    zs = power_vector(z, q_len + 1)[1:]
    zw_l = vector_matrix_product(zs, w_l)
    zw_r = vector_matrix_product(zs, w_r)
    zw_o = vector_matrix_product(zs, w_o)
    hsp = [point_mul(hi, yi) for hi, yi in zip(hs(n), power_vector(pow(y, -1, R), n))]
    u = point_mul(GEN, shamir_u(proof["tBlinding"], proof["mu"], proof["t"]))
    # verifyTPoly (:43-54)
    lhs = commit(proof["t"], proof["tBlinding"])
    k = ac_delta(n, y, zw_l, zw_r)
    c_q = dot(zs, [v % R for v in c["cs"]])
    g_exp = x * x * (k + c_q) % R
    v_exp = [x * x * v % R for v in vector_matrix_product(zs, c["commitmentWeights"])]
    xs = [0, x, 0] + [pow(x, e, R) for e in range(3, 7)]
    rhs = point_add(point_add(point_mul(GEN, g_exp), sum_exps(xs, proof["tCommits"])), sum_exps(v_exp, v_commits))
    verify_t_poly_ok = lhs == rhs
    # verifyLRCommitment (:56-71)
    commitment_lr = ac_lr_commitment(n, (proof["aiCommit"], proof["aoCommit"], proof["sCommit"]), x, y,
                                     zw_l, zw_r, zw_o, proof["mu"], proof["t"], u, hsp)
    lr_ok = ipp_verify_proof(n, gs(n), hsp, u, commitment_lr, proof["productProof"])
    return lr_ok and verify_t_poly_ok


def ac_draw_blinders(rng, n):
    """Test stand-in for MonadRandom in the AC prover's draw order and ranges
    (ArithmeticCircuit/Prover.hs:24-27,32,57; Utils.hs:135-139).  n = padded gate count."""
    three = [rng.randint(1, R - 1) for _ in range(3)]
    s_l = [rng.randint(1, 2**n - 1) % R for _ in range(n)] if n else []
    s_r = [rng.randint(1, 2**n - 1) % R for _ in range(n)] if n else []
    five = [rng.randint(1, R - 1) for _ in range(5)]
    return (*three, s_l, s_r, five)


# --------------------------------------------------------------------------
# Proof bytes and `show` text.  NOT part of the reference (it has no serialisation, only derived
# Show/Eq/Generic/NFData: RangeProof/Internal.hs:38, InnerProductProof/Internal.hs:24); this is the
# independent restatement of the layout documented in include/bpgpu.h, used to cross-check
# bp_rangeproof_serialize / _deserialize / _show.
# --------------------------------------------------------------------------
def wire_point(pt):
    return b"\x00" if pt is O else b"\x04" + pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def wire_encode(proof):
    ipp = proof["productProof"]
    out = b"".join(proof[k].to_bytes(32, "big") for k in ("tBlinding", "mu", "t"))
    out += b"".join(wire_point(proof[k]) for k in ("aCommit", "sCommit", "t1Commit", "t2Commit"))
    out += bytes([len(ipp["lCommits"])])
    out += b"".join(wire_point(p) for p in ipp["lCommits"]) + b"".join(wire_point(p) for p in ipp["rCommits"])
    return out + ipp["l"].to_bytes(32, "big") + ipp["r"].to_bytes(32, "big")


def show_range_proof(proof):
    """Derived `Show` of the record types (record syntax, declaration order of the fields)."""
    ipp = proof["productProof"]
    lst = lambda ps: b"[" + b",".join(show_point(p) for p in ps) + b"]"
    return (b"RangeProof {tBlinding = " + show_fr(proof["tBlinding"]) + b", mu = " + show_fr(proof["mu"])
            + b", t = " + show_fr(proof["t"]) + b", aCommit = " + show_point(proof["aCommit"])
            + b", sCommit = " + show_point(proof["sCommit"]) + b", t1Commit = " + show_point(proof["t1Commit"])
            + b", t2Commit = " + show_point(proof["t2Commit"])
            + b", productProof = InnerProductProof {lCommits = " + lst(ipp["lCommits"]) + b", rCommits = "
            + lst(ipp["rCommits"]) + b", l = " + show_fr(ipp["l"]) + b", r = " + show_fr(ipp["r"]) + b"}}")
