"""Quick probe of the batch verifier: replicates the golden n=64 proof to B proofs."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from bulletproofs_amd import Engine, _lib
from util import golden, up, load_proof

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--chunks", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
eng = Engine(0)
t0 = time.perf_counter(); eng.derive_generators(64); print("derive gens s", time.perf_counter() - t0, flush=True)
b = golden()["bench_proof"]; pr = load_proof(b["proof"]); vc = up(b["vCommit"]); ipp = pr["productProof"]
B = a.batch
sc_rows = [pr["tBlinding"], pr["mu"], pr["t"], ipp["l"], ipp["r"]]
pt_rows = [pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"]] + ipp["lCommits"] + ipp["rCommits"] + [vc]
sc = np.concatenate([np.tile(np.frombuffer(_lib.enc_scalar(s), dtype=np.uint8), B) for s in sc_rows])
pts = np.concatenate([np.tile(np.frombuffer(_lib.enc_point(p), dtype=np.uint8), B) for p in pt_rows])
# corrupt every 100th proof's t
scv = sc.reshape(5, B, 32); scv[2, ::100, 0] ^= 1
job = eng.vjob(64, 1, B, a.chunks, a.threads)
job.load(sc.tobytes(), pts.tobytes())
times = []
for it in range(a.iters):
    t0 = time.perf_counter(); v = job.run(); dt = time.perf_counter() - t0
    times.append(dt)
    ok = np.frombuffer(v, dtype=np.uint8)
    exp = np.ones(B, dtype=np.uint8); exp[::100] = 0
    print(json.dumps(dict(batch=B, chunks=a.chunks, threads=a.threads, s=dt, verifies_per_s=B / dt, correct=bool((ok == exp).all()))), flush=True)
ts = sorted(times[1:]) or times
print(json.dumps(dict(summary=True, iters=len(times), min_ms=ts[0] * 1e3, median_ms=ts[len(ts) // 2] * 1e3)), flush=True)
job.close()
