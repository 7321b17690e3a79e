"""Steady-state probe of the batch verifier: N passes over the same 65,536 proofs, synchronous (one job) or with two
(mode "pipe") or J (mode "pipeJ", e.g. pipe3) jobs in flight (bp_vjob_submit / bp_vjob_wait).  Run under rocprofv3 --kernel-trace and feed the .db to
benchmarks/prof_window.py to see how busy the GPU is inside the passes."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from bulletproofs_amd import Engine, _lib  # noqa: E402
import proofgen  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
chunks = int(sys.argv[5]) if len(sys.argv) > 5 else 0
eng = Engine(0)
_, _, h = eng.derive_generators(64)
inp = proofgen.random_inputs(4, 64, B)
sc, pts = proofgen.make_proofs(eng, _lib.enc_point(h), 64, B, inp, threads=16)
njobs = 1 if mode == "sync" else (int(mode[4:]) if len(mode) > 4 else 2)
jobs = [eng.vjob(64, 1, B, chunks, threads) for _ in range(njobs)]
for j in jobs:
    j.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))
    v = np.frombuffer(j.run(), dtype=np.uint8)
    assert v.all()
t0 = time.perf_counter()
if njobs > 1:
    for j in jobs:
        j.submit()
    for i in range(passes - njobs):
        j = jobs[i % njobs]
        j.wait()
        j.submit()
    for i in range(passes - njobs, passes):
        jobs[i % njobs].wait()
else:
    for _ in range(passes):
        jobs[0].run()
dt = time.perf_counter() - t0
print("%s chunks=%d threads=%d: %d passes of %d proofs, %.2f ms per pass, %.3f M verifies/s"
      % (mode, chunks, threads, passes, B, dt / passes * 1e3, B * passes / dt / 1e6))
for j in jobs:
    j.close()
eng.close()
