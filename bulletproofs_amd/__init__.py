"""bulletproofs_amd - MI355X-native engine for the elliptic-curve hot path of
sdiehl/bulletproofs (secp256k1), behind the C ABI of include/bpgpu.h.

The Python layer is a thin ctypes driver over libbpgpu.so used by the tests and
bench.py; it mirrors the reference's function names (Bulletproofs.Utils
`sumExps`, `addTwoMulP`, `commit`, ...).  There is no CPU fallback: importing
the engine without the built HIP library raises.
"""
from ._lib import Engine, load_library, BpError  # noqa: F401
