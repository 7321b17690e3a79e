"""ctypes driver of benchmarks/libbpbench.so: seeded honest range proofs (m = 1) in the
bp_vbatch_load layout.  Bench/test tooling, not part of the product path."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        from bulletproofs_amd import load_library
        load_library()  # libbpgpu.so first (and torch's HIP runtime before it)
        L = ctypes.CDLL(os.path.join(HERE, "libbpbench.so"))
        vp, u8 = ctypes.c_void_p, ctypes.c_char_p
        L.bpbench_make_proofs.restype = ctypes.c_int
        L.bpbench_make_proofs.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32] + [vp] * 9 + \
            [ctypes.c_int, vp, vp]
        L.bpbench_random_inputs.restype = None
        L.bpbench_random_inputs.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32] + [vp] * 8
        _lib = L
    return _lib


def random_inputs(seed, n_bits, batch):
    """dict of numpy uint8 arrays: v, gamma, alpha, rho, tau1, tau2 [B,32]; sL, sR [B,n,32]."""
    a = {k: np.zeros((batch, 32), dtype=np.uint8) for k in ("v", "gamma", "alpha", "rho", "tau1", "tau2")}
    a["sL"] = np.zeros((batch, n_bits, 32), dtype=np.uint8)
    a["sR"] = np.zeros((batch, n_bits, 32), dtype=np.uint8)
    p = lambda k: a[k].ctypes.data
    lib().bpbench_random_inputs(seed, n_bits, batch, p("v"), p("gamma"), p("sL"), p("sR"), p("alpha"), p("rho"),
                                p("tau1"), p("tau2"))
    return a


def make_proofs(eng, h_point_bytes, n_bits, batch, inputs, threads=8, show_style=0):
    """-> (scalars [5,B,32], points [4+2k+1,B,64]) numpy uint8 arrays."""
    k = n_bits.bit_length() - 1
    sc = np.zeros((5, batch, 32), dtype=np.uint8)
    pts = np.zeros((4 + 2 * k + 1, batch, 64), dtype=np.uint8)
    hbuf = np.frombuffer(h_point_bytes, dtype=np.uint8).copy()
    p = lambda k_: inputs[k_].ctypes.data
    rc = lib().bpbench_make_proofs(eng.h, show_style, n_bits, batch, hbuf.ctypes.data, p("v"), p("gamma"), p("sL"),
                                   p("sR"), p("alpha"), p("rho"), p("tau1"), p("tau2"), threads, sc.ctypes.data,
                                   pts.ctypes.data)
    if rc != 0:
        raise RuntimeError("bpbench_make_proofs failed: %d" % rc)
    return sc, pts
