"""ctypes binding of include/bpgpu.h.  Fails loudly when libbpgpu.so is absent."""
import ctypes
import os

# The batch verifier overlaps its chunks on streams that share a hardware pipe; HIP's default of 4
# hardware queues per process leaves no such pair ("streams").  Must be set before the HIP
# runtime initialises.  Values below 8 (HIP's default is 4) are raised to 16 unless BP_KEEP_HW_QUEUES is set.
def _want_hw_queues():
    if os.environ.get("BP_KEEP_HW_QUEUES"):
        return
    try:
        cur = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        cur = 4
    if cur < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"


_want_hw_queues()

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbpgpu.so")

P = 2**256 - 2**32 - 977
R = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


class BpError(RuntimeError):
    pass


ERR_REFERENCE_FAILS = -8  # BP_ERR_REFERENCE_FAILS


_lib = None

# name -> (restype, argtypes); every symbol include/bpgpu.h declares
_u8p = ctypes.c_char_p
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_int = ctypes.c_int
SYMBOLS = {
    "bp_version": (_int, []),
    "bp_ctx_create": (_int, [ctypes.POINTER(_vp), _int]),
    "bp_ctx_destroy": (None, [_vp]),
    "bp_last_error": (ctypes.c_char_p, [_vp]),
    "bp_ctx_sync": (_int, [_vp]),
    "bp_ctx_stream": (_vp, [_vp]),
    "bp_ctx_queue_info": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "bp_ctx_profile": (_int, [_vp, _int]),
    "bp_ctx_profile_read": (_int, [_vp, _int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                   ctypes.POINTER(ctypes.c_double)]),
    "bp_point_mul": (_int, [_vp, _u8p, _u8p, _sz, _u8p]),
    "bp_point_add": (_int, [_vp, _u8p, _u8p, _sz, _u8p]),
    "bp_point_neg": (_int, [_vp, _u8p, _sz, _u8p]),
    "bp_add_two_mul": (_int, [_vp, _u8p, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bp_point_on_curve": (_int, [_vp, _u8p, _sz, _u8p]),
    "bp_msm": (_int, [_vp, _u8p, _u8p, _sz, _u8p]),
    "bp_msm_dev": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "bp_msm_dev_hs": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "bp_msm_points_create": (_int, [_vp, _u8p, _sz, ctypes.POINTER(_vp)]),
    "bp_msm_points_create_dev": (_int, [_vp, _vp, _sz, ctypes.POINTER(_vp)]),
    "bp_msm_points_destroy": (None, [_vp]),
    "bp_msm_points_run": (_int, [_vp, _vp, _vp, _u8p]),
    "bp_msm_points_run_dev": (_int, [_vp, _vp, _vp, _vp]),
    "bp_point_sum_dev": (_int, [_vp, _vp, _sz, _vp]),
    "bp_msm_dev_static": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "bp_field_op": (_int, [_vp, _int, _int, _u8p, _u8p, _sz, _u8p]),
    "bp_fr_vector_op": (_int, [_vp, _int, _u8p, _u8p, ctypes.c_uint32, _u8p]),
    "bp_rp_polys_selftest": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32] + [_u8p] * 7),
    "bp_ipp_fold_selftest": (_int, [_vp, ctypes.c_uint32, _u8p, _u8p, _u8p, _u8p, ctypes.c_uint32, _u8p, _u8p]),
    "bp_show_numbers": (_int, [_vp, _u8p, _sz, _u8p]),
    "bp_generators_save": (_int, [_vp, ctypes.c_char_p]),
    "bp_generators_load": (_int, [_vp, ctypes.c_char_p]),
    "bp_ctx_set_generators": (_int, [_vp, _u8p, _u8p, _u8p, ctypes.c_uint32]),
    "bp_vbatch_create": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "bp_vbatch_destroy": (None, [_vp]),
    "bp_vbatch_rounds": (ctypes.c_uint32, [_vp]),
    "bp_vbatch_point_rows": (ctypes.c_uint32, [_vp]),
    "bp_vbatch_load": (_int, [_vp, _u8p, _u8p]),
    "bp_vbatch_begin": (_int, [_vp, _u8p, _u8p]),
    "bp_vbatch_round": (_int, [_vp, ctypes.c_uint32, _u8p, _u8p]),
    "bp_vbatch_finish": (_int, [_vp, _u8p]),
    "bp_ctx_set_show_style": (_int, [_vp, _int]),
    "bp_ctx_set_sqrt_branch": (_int, [_vp, _int]),
    "bp_generators_derive": (_int, [_vp, ctypes.c_uint32, _u8p, _u8p, _u8p]),
    "bp_log_base2": (_int, [_u8p, _sz]),
    "bp_transcript": (_int, [_int, _int, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bp_rangeproof_verify_batch": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _u8p, _u8p, _u8p]),
    "bp_rangeproof_verify": (_int, [_vp, _u8p, _sz, _u8p, ctypes.c_uint32, _u8p, _u8p, ctypes.c_uint32, _u8p]),
    "bp_vjob_create": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _int,
                              ctypes.POINTER(_vp)]),
    "bp_vjob_destroy": (None, [_vp]),
    "bp_vjob_load": (_int, [_vp, _u8p, _u8p]),
    "bp_vjob_run": (_int, [_vp, _u8p]),
    "bp_vjob_submit": (_int, [_vp]),
    "bp_vjob_wait": (_int, [_vp, _u8p]),
    "bp_ipp_prove": (_int, [_vp, ctypes.c_uint32] + [_u8p] * 10),
    "bp_ipp_verify": (_int, [_vp, ctypes.c_uint64, _u8p, ctypes.c_uint32, _u8p, ctypes.c_uint32, _u8p, _u8p, _u8p, _u8p,
                             ctypes.c_uint32, _u8p, _u8p, _u8p]),
    "bp_upper_bound_bits": (_int, [_u8p, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "bp_rangeproof_prove": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32] + [_u8p] * 8 + [_int, _u8p, _u8p]),
    "bp_rangeproof_wire_size": (_sz, [ctypes.c_uint32]),
    "bp_rangeproof_serialize": (_int, [_u8p, _u8p, ctypes.c_uint32, _u8p, _sz, ctypes.POINTER(_sz)]),
    "bp_rangeproof_deserialize": (_int, [_u8p, _sz, ctypes.POINTER(ctypes.c_uint32), _u8p, _u8p, _sz]),
    "bp_rangeproof_show": (_int, [_u8p, _u8p, ctypes.c_uint32, _int, _u8p, _sz, ctypes.POINTER(_sz)]),
    "bp_ac_proof_points": (_int, [ctypes.c_uint32]),
    "bp_ac_prove": (_int, [_vp] + [ctypes.c_uint32] * 3 + [_u8p] * 14),
    "bp_ac_verify": (_int, [_vp] + [ctypes.c_uint32] * 3 + [_u8p] * 9),
}


def load_library():
    """dlopen libbpgpu.so and type every exported symbol.  Raises if the HIP
    extension has not been built (run `make` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime (same soname, libamdhip64.so.7).  Two HIP
    # runtimes in one process cannot both open the GPU, so when torch is around
    # let it load first: libbpgpu.so then binds to the runtime torch brought.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing for tests/bench only; the engine does not need it
        pass
    if not os.path.exists(LIB_PATH):
        raise BpError("HIP engine not built: %s is missing (run `make`); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# ---- encodings -----------------------------------------------------------------
R_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141  # _r


def enc_scalar(s):
    return int(s).to_bytes(32, "little")


def enc_fr(ss):
    """Fr literals as Haskell reads them (`fromInteger`: -1 is r - 1)."""
    return b"".join(enc_scalar(int(s) % R_ORDER) for s in ss)


def dec_scalar(b):
    return int.from_bytes(b, "little")


def enc_point(pt):
    """(x, y) or None (identity `O`) -> 64 bytes."""
    if pt is None:
        return bytes(64)
    return int(pt[0]).to_bytes(32, "little") + int(pt[1]).to_bytes(32, "little")


def dec_point(b):
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:64], "little")
    return None if (x == 0 and y == 0) else (x, y)


def enc_scalars(ss):
    return b"".join(enc_scalar(s) for s in ss)


def enc_points(ps):
    return b"".join(enc_point(p) for p in ps)


def dec_points(buf, n):
    return [dec_point(buf[64 * i: 64 * i + 64]) for i in range(n)]


def dec_scalars(buf, n):
    return [dec_scalar(buf[32 * i: 32 * i + 32]) for i in range(n)]


class Engine:
    """One engine context (one GPU, one stream).  Method names follow the
    reference: sum_exps = `sumExps`, add_two_mul_p = `addTwoMulP`
    (Bulletproofs/Utils.hs:72-80), mul/add/inv = the curve operations."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.bp_ctx_create(ctypes.byref(h), device)
        if rc != 0:
            raise BpError("bp_ctx_create(device=%d) failed: %d (no usable GPU?)" % (device, rc))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.bp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            e = BpError("%s failed (%d): %s" % (what, rc, self.lib.bp_last_error(self.h).decode()))
            e.code = rc
            raise e

    def profile(self, enable=True):
        self._check(self.lib.bp_ctx_profile(self.h, 1 if enable else 0), "bp_ctx_profile")

    def profile_read(self, kernel):
        """-> (total_ms, launches, mixed_adds) for kernel 0 (k_fixed_multiexp) or 1 (k_light)."""
        ms, n, madds = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self.lib.bp_ctx_profile_read(self.h, kernel, ctypes.byref(ms), ctypes.byref(n),
                                                 ctypes.byref(madds)), "bp_ctx_profile_read")
        return ms.value, n.value, madds.value

    def sync(self):
        self._check(self.lib.bp_ctx_sync(self.h), "bp_ctx_sync")

    # -- raw byte-level calls ---------------------------------------------------
    def point_mul_raw(self, scalars, points, n):
        out = ctypes.create_string_buffer(64 * n)
        self._check(self.lib.bp_point_mul(self.h, scalars, points, n, out), "bp_point_mul")
        return out.raw

    def add_raw(self, a, b, n):
        out = ctypes.create_string_buffer(64 * n)
        self._check(self.lib.bp_point_add(self.h, a, b, n, out), "bp_point_add")
        return out.raw

    def msm_raw(self, scalars, points, n):
        out = ctypes.create_string_buffer(64)
        self._check(self.lib.bp_msm(self.h, scalars, points, n, out), "bp_msm")
        return out.raw

    def msm_dev(self, d_scalars, d_points, n, d_out, window_bits=0):
        self._check(self.lib.bp_msm_dev(self.h, d_scalars, d_points, n, window_bits, d_out), "bp_msm_dev")

    def msm_dev_hs(self, h_scalars, d_points, n, d_out, window_bits=0):
        """bp_msm_dev_hs: scalars at a HOST address (e.g. pinned memory), points and result in device memory."""
        self._check(self.lib.bp_msm_dev_hs(self.h, h_scalars, d_points, n, window_bits, d_out), "bp_msm_dev_hs")

    def msm_points(self, points=None, d_points=None, n=None):
        """bp_msm_points_create(_dev): prepare a fixed point list (bytes, or a device pointer with n) once."""
        return MsmPoints(self, points, d_points, n)

    def point_sum_dev(self, d_points, n, d_out):
        self._check(self.lib.bp_point_sum_dev(self.h, d_points, n, d_out), "bp_point_sum_dev")

    def msm_dev_static(self, d_scalars, d_points, n, d_out, window_bits=0):
        """bp_msm_dev over the library's static scratch array: no allocation (n <= 65,536)."""
        self._check(self.lib.bp_msm_dev_static(self.h, d_scalars, d_points, n, window_bits, d_out),
                    "bp_msm_dev_static")

    def queue_info(self):
        """-> (GPU_MAX_HW_QUEUES of the process or 0, pool streams, streams measured to run concurrently)."""
        a, b, c = _int(), _int(), _int()
        self._check(self.lib.bp_ctx_queue_info(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                    "bp_ctx_queue_info")
        return a.value, b.value, c.value

    # -- list-of-ints conveniences (tests) -----------------------------------------
    def mul(self, points, scalars):
        n = len(points)
        return dec_points(self.point_mul_raw(enc_scalars(scalars), enc_points(points), n), n)

    def add(self, a, b):
        n = len(a)
        out = ctypes.create_string_buffer(64 * n)
        self._check(self.lib.bp_point_add(self.h, enc_points(a), enc_points(b), n, out), "bp_point_add")
        return dec_points(out.raw, n)

    def inv(self, a):
        n = len(a)
        out = ctypes.create_string_buffer(64 * n)
        self._check(self.lib.bp_point_neg(self.h, enc_points(a), n, out), "bp_point_neg")
        return dec_points(out.raw, n)

    def add_two_mul_p(self, e0, p0, e1, p1):
        n = len(e0)
        out = ctypes.create_string_buffer(64 * n)
        self._check(self.lib.bp_add_two_mul(self.h, enc_scalars(e0), enc_points(p0), enc_scalars(e1),
                                            enc_points(p1), n, out), "bp_add_two_mul")
        return dec_points(out.raw, n)

    def on_curve(self, pts):
        n = len(pts)
        out = ctypes.create_string_buffer(n)
        self._check(self.lib.bp_point_on_curve(self.h, enc_points(pts), n, out), "bp_point_on_curve")
        return [b != 0 for b in out.raw]

    def sum_exps(self, scalars, points):
        n = min(len(scalars), len(points))  # Utils.hs:79 truncation
        return dec_point(self.msm_raw(enc_scalars(scalars[:n]), enc_points(points[:n]), n))

    def field_op(self, field, op, a, b=None):
        n = len(a)
        if b is None:
            b = [0] * n
        out = ctypes.create_string_buffer(32 * n)
        self._check(self.lib.bp_field_op(self.h, field, op, enc_scalars(a), enc_scalars(b), n, out), "bp_field_op")
        return dec_scalars(out.raw, n)

    def fr_vector_op(self, op, a, b=None):
        """bp_fr_vector_op: 0 powerVector (a = (scalar, n)), 1 hadamard, 2 dot."""
        if op == 0:
            a, n = a
            out = ctypes.create_string_buffer(32 * max(n, 1))
            self._check(self.lib.bp_fr_vector_op(self.h, 0, enc_scalar(a), None, n, out), "bp_fr_vector_op")
            return dec_scalars(out.raw, n)
        n = len(a)
        out = ctypes.create_string_buffer(32 * max(n, 1))
        self._check(self.lib.bp_fr_vector_op(self.h, op, enc_scalars(a), enc_scalars(b), n, out), "bp_fr_vector_op")
        return dec_scalars(out.raw, n) if op == 1 else dec_scalar(out.raw[:32])

    def rp_polys(self, n_bits, m, a_l, s_l, s_r, y, z, x):
        """-> (l0, r0, r1, ls, rs, (t0, t1, t2)) from the range prover's kernels."""
        nm = n_bits * m
        out = ctypes.create_string_buffer(32 * (5 * nm + 3))
        self._check(self.lib.bp_rp_polys_selftest(self.h, n_bits, m, bytes(a_l), enc_scalars(s_l), enc_scalars(s_r),
                                                  enc_scalar(y), enc_scalar(z), enc_scalar(x), out),
                    "bp_rp_polys_selftest")
        v = dec_scalars(out.raw, 5 * nm + 3)
        return tuple(v[i * nm:(i + 1) * nm] for i in range(5)) + (tuple(v[5 * nm:]),)

    def ipp_fold(self, ls, rs, hscale, ub, xs):
        """-> (ls', rs', dot, ub*cL, ub*cR, [3][2n+1] scalars) after len(xs) rounds of the prover's witness kernels."""
        n, r = len(ls), len(xs)
        nk = n >> r
        out = ctypes.create_string_buffer(32 * (2 * nk + 3 + 3 * (2 * n + 1)))
        self._check(self.lib.bp_ipp_fold_selftest(self.h, n, enc_scalars(ls), enc_scalars(rs), enc_scalars(hscale),
                                                  enc_scalar(ub), r, enc_scalars(xs), out), "bp_ipp_fold_selftest")
        v = dec_scalars(out.raw, 2 * nk + 3 + 3 * (2 * n + 1))
        nt = 2 * n + 1
        sc = v[2 * nk + 3:]
        return v[:nk], v[nk:2 * nk], v[2 * nk], v[2 * nk + 1], v[2 * nk + 2], [sc[q * nt:(q + 1) * nt] for q in range(3)]

    def show_numbers(self, values):
        """Decimal text of 256-bit integers as the verifier's kernels format it for the host transcript."""
        n = len(values)
        out = ctypes.create_string_buffer(80 * n)
        self._check(self.lib.bp_show_numbers(self.h, enc_scalars(values), n, out), "bp_show_numbers")
        raw = out.raw
        return [raw[80 * i + 80 - raw[80 * i]:80 * i + 80].decode() for i in range(n)]

    # -- generators and the batch verifier state machine ---------------------------------
    def set_generators(self, gs, hs, h):
        """gs, hs: lists of points (`take n gs/hs`, Utils.hs:19-25), h: point."""
        n = len(gs)
        self._check(self.lib.bp_ctx_set_generators(self.h, enc_points(gs), enc_points(hs), enc_point(h), n),
                    "bp_ctx_set_generators")

    def set_show_style(self, style):
        """0: derived Show ("A (P x) (P y)", "P s"), 1: plain ("A x y", "s") - the text the transcript hashes."""
        self._check(self.lib.bp_ctx_set_show_style(self.h, style), "bp_ctx_set_show_style")

    def vbatch(self, n_bits, m, batch):
        return VBatch(self, n_bits, m, batch)

    def derive_generators(self, n):
        """`h`, `take n gs`, `take n hs` (Utils.hs:14-25) evaluated by the engine and installed."""
        g = ctypes.create_string_buffer(64 * n)
        hh = ctypes.create_string_buffer(64 * n)
        h1 = ctypes.create_string_buffer(64)
        self._check(self.lib.bp_generators_derive(self.h, n, g, hh, h1), "bp_generators_derive")
        return dec_points(g.raw, n), dec_points(hh.raw, n), dec_point(h1.raw)

    def save_generators(self, path):
        """Write the installed `h`, `gs`, `hs` to a file (bp_generators_save)."""
        self._check(self.lib.bp_generators_save(self.h, os.fsencode(path)), "bp_generators_save")

    def load_generators(self, path):
        """Install generators from a file written by save_generators (checked: checksum, on-curve)."""
        self._check(self.lib.bp_generators_load(self.h, os.fsencode(path)), "bp_generators_load")

    def verify_batch(self, n_bits, m, batch, scalars, points):
        """`map (verifyProof upperBound)` over a batch (MultiRangeProof/Verifier.hs:19-39)."""
        out = ctypes.create_string_buffer(batch)
        self._check(self.lib.bp_rangeproof_verify_batch(self.h, n_bits, m, batch, scalars, points, out),
                    "bp_rangeproof_verify_batch")
        return [b != 0 for b in out.raw]

    def verify_proof(self, upper_bound, v_commits, proof):
        """MultiRangeProof.verifyProof upperBound vCommits proof (MultiRangeProof/Verifier.hs:19-39) for one proof,
        total like the reference: any bound, any number of L/R pairs.  Raises BpError with .code =
        ERR_REFERENCE_FAILS where the reference raises."""
        nbytes = max(1, (upper_bound.bit_length() + 7) // 8)
        ipp = proof["productProof"]
        k = min(len(ipp["lCommits"]), len(ipp["rCommits"]))  # zip
        sc = enc_scalars([proof["tBlinding"], proof["mu"], proof["t"], ipp["l"], ipp["r"]])
        pts = enc_points([proof["aCommit"], proof["sCommit"], proof["t1Commit"], proof["t2Commit"]]
                         + list(ipp["lCommits"])[:k] + list(ipp["rCommits"])[:k])
        out = ctypes.create_string_buffer(1)
        self._check(self.lib.bp_rangeproof_verify(self.h, upper_bound.to_bytes(nbytes, "little"), nbytes,
                                                  enc_points(v_commits), len(v_commits), sc, pts, k, out),
                    "bp_rangeproof_verify")
        return out.raw[0] != 0

    def ipp_prove(self, gs, hs, u, P, ls, rs):
        """IPP.generateProof (InnerProductProof/Prover.hs:17-26) -> dict(lCommits, rCommits, l, r)."""
        n = len(ls)
        k = max(n.bit_length() - 1, 0)
        L = ctypes.create_string_buffer(64 * max(k, 1))
        R_ = ctypes.create_string_buffer(64 * max(k, 1))
        l = ctypes.create_string_buffer(32)
        r = ctypes.create_string_buffer(32)
        self._check(self.lib.bp_ipp_prove(self.h, n, enc_points(gs), enc_points(hs), enc_point(u), enc_point(P),
                                          enc_scalars(ls), enc_scalars(rs), L, R_, l, r), "bp_ipp_prove")
        return dict(lCommits=dec_points(L.raw, k), rCommits=dec_points(R_.raw, k), l=dec_scalar(l.raw),
                    r=dec_scalar(r.raw))

    def ipp_verify(self, n, gs, hs, u, P, proof):
        """IPP.verifyProof n (InnerProductBase gs hs u) P proof (InnerProductProof/Verifier.hs:18-38) -> bool."""
        Ls, Rs = list(proof["lCommits"]), list(proof["rCommits"])
        k = min(len(Ls), len(Rs))  # zip
        out = ctypes.create_string_buffer(1)
        self._check(self.lib.bp_ipp_verify(self.h, n, enc_points(gs), len(gs), enc_points(hs), len(hs), enc_point(u),
                                           enc_point(P), enc_points(Ls[:k]), enc_points(Rs[:k]), k,
                                           enc_scalar(proof["l"] % R_ORDER), enc_scalar(proof["r"] % R_ORDER), out),
                    "bp_ipp_verify")
        return out.raw[0] != 0

    def prove(self, n_bits, vs_and_blindings, s_l, s_r, a_blinding, s_blinding, t1_blinding, t2_blinding,
              checked=True):
        """MultiRangeProof.generateProof / generateProofUnsafe (MultiRangeProof/Prover.hs:19-123) with the
        blinders as explicit inputs; returns the proof as the oracle's dict, or raises RangeProofError-like
        BpError carrying .code."""
        m = len(vs_and_blindings)
        m2 = 1
        while m2 < m:
            m2 *= 2
        k = (n_bits * m2).bit_length() - 1
        sc = ctypes.create_string_buffer(5 * 32)
        pts = ctypes.create_string_buffer((4 + 2 * k) * 64)
        rc = self.lib.bp_rangeproof_prove(
            self.h, n_bits, m, enc_scalars([v % (2**256) for v, _ in vs_and_blindings]),
            enc_scalars([b for _, b in vs_and_blindings]), enc_scalars(s_l), enc_scalars(s_r),
            enc_scalar(a_blinding), enc_scalar(s_blinding), enc_scalar(t1_blinding), enc_scalar(t2_blinding),
            1 if checked else 0, sc, pts)
        if rc != 0:
            e = BpError("bp_rangeproof_prove failed (%d): %s" % (rc, self.lib.bp_last_error(self.h).decode()))
            e.code = rc
            raise e
        s = dec_scalars(sc.raw, 5)
        p = dec_points(pts.raw, 4 + 2 * k)
        return dict(tBlinding=s[0], mu=s[1], t=s[2], aCommit=p[0], sCommit=p[1], t1Commit=p[2], t2Commit=p[3],
                    productProof=dict(lCommits=p[4:4 + k], rCommits=p[4 + k:4 + 2 * k], l=s[3], r=s[4]))

    @staticmethod
    def _enc_circuit(circuit):
        """dict(wL, wR, wO, commitmentWeights, cs) (ArithCircuit, ArithmeticCircuit/Internal.hs:42-58) -> dense bytes."""
        w_l = circuit["wL"]
        q = len(circuit["commitmentWeights"])
        n = len(w_l[0]) if w_l else 0
        m = len(circuit["commitmentWeights"][0]) if q else 0
        flat = lambda rows: enc_fr([v for row in rows for v in row])
        return n, q, m, flat(w_l), flat(circuit["wR"]), flat(circuit["wO"]), flat(circuit["commitmentWeights"]), \
            enc_fr(circuit["cs"])

    def ac_prove(self, circuit, witness, ai_blinding, ao_blinding, s_blinding, s_l, s_r, t_randoms):
        """ArithmeticCircuit.generateProof (ArithmeticCircuit/Prover.hs:16-117) with explicit blinders;
        returns the proof as the oracle's dict."""
        n, q, m, w_l, w_r, w_o, w_v, _cs = self._enc_circuit(circuit)
        asg = witness["assignment"]
        assert len(asg["aL"]) == n
        npts = self.lib.bp_ac_proof_points(n)
        k = (npts - 10) // 2
        sc = ctypes.create_string_buffer(5 * 32)
        pts = ctypes.create_string_buffer(npts * 64)
        self._check(self.lib.bp_ac_prove(
            self.h, n, q, m, w_l, w_r, w_o, w_v, enc_fr(asg["aL"]), enc_fr(asg["aR"]),
            enc_fr(asg["aO"]), enc_fr(witness["commitBlinders"]),
            enc_fr([ai_blinding, ao_blinding, s_blinding]), enc_fr(s_l), enc_fr(s_r),
            enc_fr(t_randoms), sc, pts), "bp_ac_prove")
        s = dec_scalars(sc.raw, 5)
        p = dec_points(pts.raw, npts)
        return dict(tBlinding=s[0], mu=s[1], t=s[2], aiCommit=p[0], aoCommit=p[1], sCommit=p[2], tCommits=p[3:10],
                    productProof=dict(lCommits=p[10:10 + k], rCommits=p[10 + k:10 + 2 * k], l=s[3], r=s[4]))

    def ac_verify(self, v_commits, proof, circuit):
        """ArithmeticCircuit.verifyProof (ArithmeticCircuit/Verifier.hs:16-71) -> bool."""
        n, q, m, w_l, w_r, w_o, w_v, cs = self._enc_circuit(circuit)
        assert len(v_commits) == m
        ipp = proof["productProof"]
        sc = enc_scalars([proof["tBlinding"], proof["mu"], proof["t"], ipp["l"], ipp["r"]])
        pts = enc_points([proof["aiCommit"], proof["aoCommit"], proof["sCommit"]] + list(proof["tCommits"])
                         + list(ipp["lCommits"]) + list(ipp["rCommits"]))
        assert len(pts) == 64 * self.lib.bp_ac_proof_points(n)
        out = ctypes.create_string_buffer(1)
        self._check(self.lib.bp_ac_verify(self.h, n, q, m, w_l, w_r, w_o, w_v, cs, enc_points(v_commits), sc, pts,
                                          out), "bp_ac_verify")
        return out.raw[0] != 0

    def vjob(self, n_bits, m, batch, n_chunks=0, host_threads=0):
        return VJob(self, n_bits, m, batch, n_chunks, host_threads)


def transcript(style, which, pts=(), scs=(), raw=b""):
    """bp_transcript: the reference's shamir*/oracle functions as the host layer computes them."""
    lib = load_library()
    out = ctypes.create_string_buffer(32)
    rc = lib.bp_transcript(style, which, enc_points(pts), enc_scalars(scs), raw, len(raw), out)
    if rc != 0:
        raise BpError("bp_transcript failed: %d" % rc)
    return dec_scalar(out.raw)


def _proof_arrays(proof):
    ipp = proof["productProof"]
    sc = enc_scalars([proof["tBlinding"], proof["mu"], proof["t"], ipp["l"], ipp["r"]])
    pts = enc_points([proof["aCommit"], proof["sCommit"], proof["t1Commit"], proof["t2Commit"]]
                     + list(ipp["lCommits"]) + list(ipp["rCommits"]))
    return sc, pts, len(ipp["lCommits"])


def serialize_proof(proof):
    """Range proof (the oracle's dict) -> canonical bytes (bp_rangeproof_serialize)."""
    lib = load_library()
    sc, pts, k = _proof_arrays(proof)
    cap = lib.bp_rangeproof_wire_size(k)
    out = ctypes.create_string_buffer(cap)
    n = _sz()
    rc = lib.bp_rangeproof_serialize(sc, pts, k, out, cap, ctypes.byref(n))
    if rc != 0:
        raise BpError("bp_rangeproof_serialize failed: %d" % rc)
    return out.raw[:n.value]


def deserialize_proof(data):
    """Canonical bytes -> range proof dict; raises BpError on any malformed input."""
    lib = load_library()
    k = ctypes.c_uint32()
    sc = ctypes.create_string_buffer(5 * 32)
    cap = 4 + 2 * 255
    pts = ctypes.create_string_buffer(cap * 64)
    rc = lib.bp_rangeproof_deserialize(bytes(data), len(data), ctypes.byref(k), sc, pts, cap)
    if rc != 0:
        raise BpError("bp_rangeproof_deserialize failed: %d" % rc)
    kk = k.value
    s = dec_scalars(sc.raw, 5)
    p = dec_points(pts.raw, 4 + 2 * kk)
    return dict(tBlinding=s[0], mu=s[1], t=s[2], aCommit=p[0], sCommit=p[1], t1Commit=p[2], t2Commit=p[3],
                productProof=dict(lCommits=p[4:4 + kk], rCommits=p[4 + kk:4 + 2 * kk], l=s[3], r=s[4]))


def show_proof(proof, style=0):
    """`show proof` as GHC derives it for RangeProof Fr PA."""
    lib = load_library()
    sc, pts, k = _proof_arrays(proof)
    cap = 1024 + (4 + 2 * k) * 192
    out = ctypes.create_string_buffer(cap)
    n = _sz()
    rc = lib.bp_rangeproof_show(sc, pts, k, style, out, cap, ctypes.byref(n))
    if rc != 0:
        raise BpError("bp_rangeproof_show failed: %d" % rc)
    return out.raw[:n.value]


def upper_bound_bits(upper_bound):
    """generateProof's validation of the upper bound -> n, or the negative error code."""
    lib = load_library()
    nbytes = max(1, (upper_bound.bit_length() + 7) // 8)
    n = ctypes.c_uint32()
    rc = lib.bp_upper_bound_bits(upper_bound.to_bytes(nbytes, "little"), nbytes, ctypes.byref(n))
    return n.value if rc == 0 else rc


def log_base2(upper_bound):
    lib = load_library()
    nbytes = max(1, (upper_bound.bit_length() + 7) // 8)
    return lib.bp_log_base2(upper_bound.to_bytes(nbytes, "little"), nbytes)


class MsmPoints:
    """bp_msm_points_*: `sumExps es ps` for a point list that stays the same call after call."""

    def __init__(self, eng, points, d_points, n):
        self.eng = eng
        h = ctypes.c_void_p()
        if points is not None:
            n = len(points) // 64
            rc = eng.lib.bp_msm_points_create(eng.h, points, n, ctypes.byref(h))
        else:
            rc = eng.lib.bp_msm_points_create_dev(eng.h, d_points, n, ctypes.byref(h))
        eng._check(rc, "bp_msm_points_create")
        self.h, self.n = h, n

    def run(self, scalars):
        """scalars: bytes (n x 32) or an int host address (e.g. of pinned memory) -> 64 result bytes."""
        out = ctypes.create_string_buffer(64)
        ptr = scalars if isinstance(scalars, int) else ctypes.cast(ctypes.c_char_p(scalars), ctypes.c_void_p)
        self.eng._check(self.eng.lib.bp_msm_points_run(self.eng.h, self.h, ptr, out), "bp_msm_points_run")
        return out.raw

    def run_dev(self, d_scalars, d_out):
        self.eng._check(self.eng.lib.bp_msm_points_run_dev(self.eng.h, self.h, d_scalars, d_out), "bp_msm_points_run_dev")

    def close(self):
        if self.h:
            self.eng.lib.bp_msm_points_destroy(self.h)
            self.h = None


class VJob:
    """bp_vjob_*: create / load (untimed) / run (timed) batch verification."""

    def __init__(self, eng, n_bits, m, batch, n_chunks, host_threads):
        self.eng = eng
        self.B = batch
        h = ctypes.c_void_p()
        eng._check(eng.lib.bp_vjob_create(eng.h, n_bits, m, batch, n_chunks, host_threads, ctypes.byref(h)),
                   "bp_vjob_create")
        self.h = h

    def load(self, scalars, points):
        self.eng._check(self.eng.lib.bp_vjob_load(self.h, scalars, points), "bp_vjob_load")

    def run(self):
        out = ctypes.create_string_buffer(self.B)
        self.eng._check(self.eng.lib.bp_vjob_run(self.h, out), "bp_vjob_run")
        return out.raw

    def submit(self):
        """Start the pass on the job's driver thread (bp_vjob_submit); collect it with wait()."""
        self.eng._check(self.eng.lib.bp_vjob_submit(self.h), "bp_vjob_submit")

    def wait(self):
        out = ctypes.create_string_buffer(self.B)
        self.eng._check(self.eng.lib.bp_vjob_wait(self.h, out), "bp_vjob_wait")
        return out.raw

    def close(self):
        if self.h:
            self.eng.lib.bp_vjob_destroy(self.h)
            self.h = None


class VBatch:
    """bp_vbatch_* state machine: B proofs of one shape in lock-step; the caller
    supplies the Fiat-Shamir challenges."""

    def __init__(self, eng, n_bits, m, batch):
        self.eng = eng
        self.B = batch
        h = ctypes.c_void_p()
        eng._check(eng.lib.bp_vbatch_create(eng.h, n_bits, m, batch, ctypes.byref(h)), "bp_vbatch_create")
        self.h = h
        self.rounds = eng.lib.bp_vbatch_rounds(h)
        self.point_rows = eng.lib.bp_vbatch_point_rows(h)

    def close(self):
        if self.h:
            self.eng.lib.bp_vbatch_destroy(self.h)
            self.h = None

    def load(self, scalars, points):
        """scalars: bytes 5*B*32 (rows tBlinding, mu, t, l, r); points: bytes rows*B*64."""
        assert len(scalars) == 5 * self.B * 32 and len(points) == self.point_rows * self.B * 64
        self.eng._check(self.eng.lib.bp_vbatch_load(self.h, scalars, points), "bp_vbatch_load")

    def begin(self, challenges):
        out = ctypes.create_string_buffer(64 * self.B)
        self.eng._check(self.eng.lib.bp_vbatch_begin(self.h, challenges, out), "bp_vbatch_begin")
        return out.raw

    def round(self, r, x):
        out = ctypes.create_string_buffer(64 * self.B)
        self.eng._check(self.eng.lib.bp_vbatch_round(self.h, r, x, out), "bp_vbatch_round")
        return out.raw

    def finish(self):
        out = ctypes.create_string_buffer(self.B)
        self.eng._check(self.eng.lib.bp_vbatch_finish(self.h, out), "bp_vbatch_finish")
        return [b != 0 for b in out.raw]
