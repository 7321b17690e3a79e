"""CPU test: libbpgpu.so loads and exports every symbol include/bpgpu.h declares
(no compute calls; there is no GPU here)."""
import ctypes
import os
import re

from bulletproofs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libbpgpu.so does not export %s" % n
    # and the Python binding types exactly the declared set
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_argument_errors_without_gpu():
    lib = _lib.load_library()
    assert lib.bp_version() >= 100
    assert lib.bp_ctx_create(None, 0) == -1  # BP_ERR_ARG: null out pointer
    assert lib.bp_last_error(None) == b"null context"
    assert lib.bp_msm(None, None, None, 0, None) == -1


def test_upper_bound_validation_needs_no_gpu():
    """generateProof's guards (MultiRangeProof/Prover.hs:25-38) are host arithmetic."""
    from bulletproofs_amd import _lib
    import bp_oracle as o
    assert _lib.upper_bound_bits(2**64) == 64
    assert _lib.upper_bound_bits(2**2) == 2
    assert _lib.upper_bound_bits(o.R) == -6
    assert _lib.upper_bound_bits(o.R - 1) == -7
    assert _lib.upper_bound_bits(2**3) == -7
    assert _lib.upper_bound_bits(2**0) == -7      # n = 0: logBase2M 0 = Nothing
    assert _lib.upper_bound_bits(0) == -7
