#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X engine (BASELINE.json metric:
"range-proof verifies/sec (64-bit, batched) at 1/2/4/8 GPU; MSM points/sec").

One step = one pass of the batch verifier over one batch of synthetic proofs: config 4 of BASELINE.json, 65,536
independent 64-bit range proofs (m = 1), 1 % of them corrupted.  The proofs are resident in HBM (and in pinned host
memory for the host-side Fiat-Shamir hashing, which is part of the timed region) before the clock starts.  Per-proof
verdicts are checked against the expected ones after every run, and at N = 1 against the C restatement of the
reference on a sample.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: proofs are independent, so ranks share nothing on the data path (only the timing barrier).  The headline
`value` is WEAK scaling (every GPU verifies its own 65,536 proofs: per-GPU work is fixed, as the bench contract
prescribes for a path that shards without a collective); `--scaling strong` makes the headline BASELINE config 4 as
literally stated (the 65,536 proofs SPLIT over the N GPUs), and with the default the same split is measured next to the
headline and reported under "strong_split".  The MSM leg (config 3) shards the point range over the ranks, keeps every
shard resident in HBM and all-gathers the 64-byte partial sums over RCCL.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))

# more hardware queues than HIP's default of 4, before anything initialises the HIP runtime: the verifier's
# chunk streams only overlap when they share a hardware pipe (bulletproofs_amd/csrc/ctx.hip).  The library itself
# never touches the environment; this harness does, unless BP_KEEP_HW_QUEUES is set.
if not os.environ.get("BP_KEEP_HW_QUEUES") and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bulletproofs_amd import Engine, _lib  # noqa: E402
from bulletproofs_amd import dist as bpdist  # noqa: E402
import proofgen  # noqa: E402

# gfx950 integer-VALU roofline: v_mad_u64_u32 issues at 16 lanes/clk/SIMD (measured 4.6 cycles per
# wave64 instruction, profiles/r01_microbench_int_valu.jsonl): 256 CU x 4 SIMD x 16 x 2.4 GHz
PEAK_TMAD = 256 * 4 * 16 * 2.4e9 / 1e12
MADS_PER_MODMUL = 136   # 8-limb Montgomery CIOS, 2*8^2 + 8
MODMUL_PER_MADD = 10    # XYZZ mixed addition, 8M + 2S
MODMUL_PER_ADD = 14     # full addition
MODMUL_PER_DBL = 10     # doubling
GLV_BITS = 129
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def corrupt(sc, pts, k, rng):
    """Flip t / swap L0,R0 / replace V on 1 % of the proofs; returns the expected verdicts."""
    B = sc.shape[1]
    bad = rng.choice(B, size=max(1, B // 100), replace=False)
    exp = np.ones(B, dtype=np.uint8)
    exp[bad] = 0
    for j, p in enumerate(bad):
        kind = j % 3
        if kind == 0:
            sc[2, p, 0] ^= 1
        elif kind == 1:
            tmp = pts[4, p].copy()
            pts[4, p] = pts[4 + k, p]
            pts[4 + k, p] = tmp
        else:
            pts[4 + 2 * k, p] = pts[4 + 2 * k, (p + 1) % B]
    return exp


def cpu_baseline(sc, pts, n_bits, sample, gpu_verdicts):
    """The reference's CPU algorithm (oracle/bp_ref.c, plain C port, 1 core) on the first
    `sample` proofs of the same batch; also cross-checks the GPU verdicts."""
    so = os.path.join(ROOT, "oracle", "libbp_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = ctypes.CDLL(so)
    L.bpref_init(0, 0, n_bits)
    B = sc.shape[1]
    out = ctypes.create_string_buffer(sample)
    t0 = time.perf_counter()
    L.bpref_verify_batch(n_bits, 1, ctypes.c_size_t(B), sc.ctypes.data_as(ctypes.c_char_p),
                         pts.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(0), ctypes.c_size_t(sample), out)
    dt = time.perf_counter() - t0
    cpu = np.frombuffer(out.raw, dtype=np.uint8)
    if not (cpu == gpu_verdicts[:sample]).all():
        raise SystemExit("PARITY FAILURE: GPU verdicts differ from the CPU restatement on the sample")
    res = dict(value=sample / dt, unit="verifies/s", cores=1, kind="port",
               sample="first %d proofs of the same batch, oracle/bp_ref.c (C restatement of the reference's "
                      "algorithm; GHC is not available), verdicts identical to the GPU's" % sample,
               algorithm="the reference's operation sequence with every `mul` in JACOBIAN coordinates (one inversion "
                         "per product) on 4x64-bit limbs: faster than the reference's own affine ladder on GMP "
                         "Integers, i.e. this baseline flatters the CPU; see affine_ladder for the literal form")
    # the LITERAL algorithm of `mul` on a `PA`: affine double-and-add, one field inversion per doubling / addition
    lit = max(4, sample // 12)
    out2 = ctypes.create_string_buffer(lit)
    L.bpref_set_mul_mode(1)
    t0 = time.perf_counter()
    L.bpref_verify_batch(n_bits, 1, ctypes.c_size_t(B), sc.ctypes.data_as(ctypes.c_char_p),
                         pts.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(0), ctypes.c_size_t(lit), out2)
    dta = time.perf_counter() - t0
    L.bpref_set_mul_mode(0)
    if not (np.frombuffer(out2.raw, dtype=np.uint8) == gpu_verdicts[:lit]).all():
        raise SystemExit("PARITY FAILURE: GPU verdicts differ from the affine-ladder CPU restatement")
    res["affine_ladder"] = dict(value=lit / dta, unit="verifies/s", cores=1, kind="port",
                                sample="first %d proofs, the same C restatement with `mul` as the literal affine "
                                       "double-and-add of Data.Curve.Weierstrass (an inversion per step, binary "
                                       "extended Euclid)" % lit)
    # the same port over the box's CPU share (independent proofs, one slice per thread; the reference itself is
    # single-threaded): a second, short figure next to the one-core baseline
    cores = min(16, os.cpu_count() or 1)
    per = max(1, min(12, sample // cores))
    if cores > 1:
        import threading
        outs = [ctypes.create_string_buffer(per) for _ in range(cores)]

        def work(t):
            L.bpref_verify_batch(n_bits, 1, ctypes.c_size_t(B), sc.ctypes.data_as(ctypes.c_char_p),
                                 pts.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(t * per), ctypes.c_size_t(per),
                                 outs[t])
        ths = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dtm = time.perf_counter() - t0
        multi = np.concatenate([np.frombuffer(o_.raw, dtype=np.uint8) for o_ in outs])
        if not (multi == gpu_verdicts[:cores * per]).all():
            raise SystemExit("PARITY FAILURE: GPU verdicts differ from the threaded CPU restatement")
        res["multi"] = dict(value=cores * per / dtm, unit="verifies/s", cores=cores,
                            sample="%d threads x %d proofs of the same batch" % (cores, per))
    return res


# ---- config 3: MSM ----------------------------------------------------------------------------------------------
def msm_choose_c(n):
    """msm_core.h msm_choose_c."""
    for lim, c in ((16384, 13), (131072, 14)):
        if n < lim:
            return c
    return 16


def msm_algorithmic_mads(n_local, direct_max=0):
    """SURVEY.md 8(d) accounting for the path a local MSM of n_local terms takes.  Bucket method with GLV halves:
    every point enters twice with a 129-bit half-scalar, W = ceil(130 / c) windows of c bits:
    modmuls = 2 n W 10 (bucket accumulation) + W 2^(c-1) 2 14 (running-sum reduction) + 128 10 (window doublings).
    Direct path (only behind BP_MSM_DIRECT_MAX since the bucket tail runs on quads): one 129-step GLV chain per term."""
    if n_local <= direct_max:
        mm = n_local * (GLV_BITS * MODMUL_PER_DBL + GLV_BITS * 0.75 * MODMUL_PER_MADD)
        return mm * MADS_PER_MODMUL, dict(path="direct", steps=GLV_BITS)
    c = msm_choose_c(n_local)
    W = (GLV_BITS + 1 + c - 1) // c
    mm = 2 * n_local * W * MODMUL_PER_MADD + W * (1 << (c - 1)) * 2 * MODMUL_PER_ADD + 128 * MODMUL_PER_DBL
    return mm * MADS_PER_MODMUL, dict(path="bucket", c=c, windows=W, glv=True)


def msm_cpu_baseline(es, pts, got):
    """`sumExps` on one host core (oracle/bp_ref.c, Utils.hs:76-80 as written: one double-and-add product per term,
    pairwise sums) over the first 2^12 / 2^9 terms of the same scalars and points, Jacobian and literal affine `mul`;
    `got(n)` returns the engine's result for the same prefix, which must be the same point."""
    so = os.path.join(ROOT, "oracle", "libbp_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = ctypes.CDLL(so)
    out = ctypes.create_string_buffer(64)
    res = {}
    for mode, n, key in ((0, 4096, "jacobian"), (1, 512, "affine_ladder")):
        n = min(n, len(es))
        L.bpref_set_mul_mode(mode)
        t0 = time.perf_counter()
        L.bpref_sum_exps(es[:n].ctypes.data_as(ctypes.c_char_p), pts[:n * 64].ctypes.data_as(ctypes.c_char_p),
                         ctypes.c_size_t(n), out)
        dt = time.perf_counter() - t0
        L.bpref_set_mul_mode(0)
        if out.raw != got(n):
            raise SystemExit("PARITY FAILURE: MSM of the first %d terms differs from the CPU restatement" % n)
        res[key] = dict(value=n / dt, unit="points/s", cores=1, kind="port",
                        sample="sumExps over the first %d terms of the 2^k-point workload, result equal to the GPU's" % n)
    base = dict(res["jacobian"])
    base["algorithm"] = "the reference's sumExps order with Jacobian products (flatters the CPU); affine_ladder = literal"
    base["affine_ladder"] = res["affine_ladder"]
    return base


def make_points(eng, n, seed):
    """n points k_i * G with seeded k_i (cheap derivation, points uploaded once)."""
    rng = np.random.default_rng(seed)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x7F
    g = np.frombuffer(_lib.enc_point((GX, GY)), dtype=np.uint8)
    pts = np.frombuffer(eng.point_mul_raw(ks.tobytes(), np.tile(g, n).tobytes(), n), dtype=np.uint8)
    return ks, pts.copy()


def make_scalars(kind, n, seed):
    rng = np.random.default_rng(seed)
    es = np.zeros((n, 32), dtype=np.uint8)
    if kind == "uniform":
        es[:] = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        es[:, 31] &= 0x7F
    elif kind == "64bit":
        es[:, :8] = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    else:  # "01": the shape commitBitVectors sees for aL
        es[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)
    return es


def msm_bench(eng, rank, world, dist, logns, iters, cpu=True):
    """MSM points/s for N = 2^logn total points sharded over the ranks by contiguous ranges (each shard resident in
    its GPU's HBM), scalars uniform / 64-bit / {0,1}.  `points_per_s` INCLUDES the upload of the scalars from pinned
    host memory every iteration and EXCLUDES the one-time point upload; the resident-scalar
    rate is reported next to it for the uniform case."""
    R = _lib.R_ORDER
    out = []
    nmax = 1 << max(logns)
    lo_max, hi_max = bpdist.shard_range(nmax, rank, world)
    ks_all, pts_all = make_points(eng, hi_max - lo_max, 3 + 100 * rank)
    d_pts_all = torch.from_numpy(pts_all).cuda()
    d_out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gathered = torch.empty((world, 64), dtype=torch.uint8, device="cuda") if world > 1 else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for logn in logns:
        n = 1 << logn
        lo, hi = bpdist.shard_range(n, rank, world)
        nl = hi - lo
        d_p = d_pts_all[:nl * 64]
        mads, how = msm_algorithmic_mads(nl)
        row = dict(n=n, n_per_gpu=nl, **how)
        for kind in ("uniform", "64bit", "01"):
            es = make_scalars(kind, nl, 30 + logn + 1000 * rank)
            h_s = torch.from_numpy(es).pin_memory()
            d_s = torch.empty_like(h_s, device="cuda")

            def once(upload):
                if upload and world == 1:
                    # scalars in pinned host memory, points resident: bp_msm_dev_hs uploads them in pieces behind
                    # which the bucket work runs (the upload is inside the timed region either way)
                    eng.msm_dev_hs(h_s.data_ptr(), d_p.data_ptr(), nl, d_out.data_ptr())
                    eng.sync()
                    return
                if upload:
                    d_s.copy_(h_s, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                bpdist.sharded_msm_dev(eng, d_s, d_p, nl, d_out, gathered=gathered)
                eng.sync()
            once(True)
            if world == 1:  # the piecewise-upload path against the resident one, every size and scalar shape
                got_hs = bytes(d_out.cpu().numpy())
                d_s.copy_(h_s)
                torch.cuda.synchronize()
                once(False)
                if bytes(d_out.cpu().numpy()) != got_hs:
                    raise SystemExit("PARITY FAILURE: bp_msm_dev_hs differs from bp_msm_dev at N = 2^%d (%s)" % (logn, kind))
            if kind == "uniform" and logn == min(logns) and n <= (1 << 17):
                # known discrete logs: sum e_i k_i * G must be the result (all ranks' shards together)
                e_int = [int.from_bytes(bytes(r_), "little") for r_ in es]
                k_int = [int.from_bytes(bytes(r_), "little") for r_ in ks_all[:nl]]
                s = sum(a * b for a, b in zip(e_int, k_int)) % R
                if dist is not None:
                    parts = [None] * world
                    dist.all_gather_object(parts, s)
                    s = sum(parts) % R
                exp = eng.mul([(GX, GY)], [s])[0]
                got = _lib.dec_point(bytes(d_out.cpu().numpy()))
                if got != exp:
                    raise SystemExit("PARITY FAILURE: MSM result differs from (sum e_i k_i) * G at N = 2^%d" % logn)
                row["checked"] = "result == (sum e_i k_i mod r) * G"
            if kind == "uniform" and logn == max(logns) and world == 1 and cpu:
                def got(m):
                    eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), m, d_out.data_ptr())
                    eng.sync()
                    return bytes(d_out.cpu().numpy())
                row["cpu_baseline"] = msm_cpu_baseline(es, pts_all, got)
            barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                once(True)
            barrier()
            dt = (time.perf_counter() - t0) / iters
            if dist is not None:
                tm = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dt = float(tm.item())
            ach = world * mads / dt / 1e12
            row[kind] = dict(ms=dt * 1e3, points_per_s=n / dt, tmad_per_s=ach, frac=ach / (world * PEAK_TMAD))
            if kind == "uniform":
                barrier()
                t0 = time.perf_counter()
                for _ in range(iters):
                    once(False)
                barrier()
                dr = (time.perf_counter() - t0) / iters
                row[kind]["resident_ms"] = dr * 1e3
                row[kind]["resident_points_per_s"] = n / dr
            if world == 1 and nl > 65536:
                # the same sum over PREPARED points (bp_msm_points: shifted copies, no window doublings), scalars
                # uploaded from pinned memory in chunks behind the recoding; result checked against the plain MSM's
                ref = bytes(d_out.cpu().numpy())
                mp = eng.msm_points(d_points=d_p.data_ptr(), n=nl)
                if mp.run(h_s.data_ptr()) != ref:
                    raise SystemExit("PARITY FAILURE: bp_msm_points_run differs from bp_msm_dev at N = 2^%d (%s)" % (logn, kind))
                t0 = time.perf_counter()
                for _ in range(iters):
                    mp.run(h_s.data_ptr())
                dp = (time.perf_counter() - t0) / iters
                row[kind]["prepared_ms"] = dp * 1e3
                row[kind]["prepared_points_per_s"] = n / dp
                row[kind]["prepared_frac"] = mads / dp / 1e12 / PEAK_TMAD
                if kind == "uniform":  # scalars resident as well: the kernels alone
                    mp.run_dev(d_s.data_ptr(), d_out.data_ptr())
                    eng.sync()
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        mp.run_dev(d_s.data_ptr(), d_out.data_ptr())
                        eng.sync()
                    dq = (time.perf_counter() - t0) / iters
                    row[kind]["prepared_resident_ms"] = dq * 1e3
                    row[kind]["prepared_resident_points_per_s"] = n / dq
                    row[kind]["prepared_resident_frac"] = mads / dq / 1e12 / PEAK_TMAD
                mp.close()
        out.append(row)
    return out


def single_proof_configs(iters):
    """BASELINE configs 1, 2, 5 (single proofs: latency) through the C ABI, host buffers in and out."""
    import contextlib
    import io
    import config_bench
    eng = Engine(torch.cuda.current_device())
    lib = eng.lib
    buf = io.StringIO()
    try:
        _, _, h = eng.derive_generators(4096)
        import random
        with contextlib.redirect_stdout(buf):
            config_bench.range_config(eng, lib, "config1_rangeproof_64", 64, [(7238283, 827361)], 1, iters, h)
            rng = random.Random(2)
            vs = [(rng.randrange(0, 2**64), rng.randrange(1, _lib.R_ORDER)) for _ in range(16)]
            config_bench.range_config(eng, lib, "config2_rangeproof_64x16", 64, vs, 2, iters, h)
            config_bench.circuit_config(eng, lib, 4096, iters, h)
    finally:
        eng.close()
    return [json.loads(line) for line in buf.getvalue().splitlines() if line.strip()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="proofs of the job (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=("auto", "strong", "weak"), default="auto",
                    help="auto = weak (the headline); the split of BASELINE config 4 is then reported as strong_split")
    ap.add_argument("--bits", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--pipe-chunks", type=int, default=0,
                    help="chunks per job of the two-jobs-in-flight leg (0: 2 with the host transcript, 1 with the device's)")
    ap.add_argument("--transcript", choices=["host", "device"], default="host",
                    help="where the headline's Fiat-Shamir transcript is formed; the other form is reported beside it")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="transcript-hashing threads per rank (0 = min(16, usable cores / ranks on this host))")
    ap.add_argument("--pin", type=int, default=1, help="with several ranks on a host: restrict each rank to its own cores")
    ap.add_argument("--cpu-sample", type=int, default=192, help="proofs of the batch timed on the CPU restatement (~16 s on one core)")
    ap.add_argument("--msm-logn", type=int, default=20, help="largest MSM size 2^k of the config-3 leg (0 = skip)")
    ap.add_argument("--msm-min-logn", type=int, default=16)
    ap.add_argument("--extras", type=int, default=1, help="single-proof configs and API-inclusive rates at N = 1")
    a = ap.parse_args()
    # Where the headline's Fiat-Shamir transcript is formed: on the host's threads (north_star's form, the library's
    # default) or on the GPU (BP_FS_DEVICE=1).  The library reads the variable when a job is created.
    fs_device = a.transcript == "device"
    os.environ["BP_FS_DEVICE"] = "1" if fs_device else "0"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (a.gpus, world, a.gpus))
    # host budget of this rank: the usable cores divided by the ranks on this host; optionally pinned to its own slice
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pinned = bpdist.pin_rank_cpus(local_rank, local_world) if (a.pin and local_world > 1) else None
    host_threads = a.host_threads or int(os.environ.get("BP_HOST_THREADS", "0")) or bpdist.host_thread_budget(local_world)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    scaling = a.scaling if a.scaling != "auto" else "weak"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng = Engine(local_rank)
    _, _, h = eng.derive_generators(a.bits)
    n_bits = a.bits
    if scaling == "strong":
        lo, hi = bpdist.shard_range(a.batch, rank, world)
        B, total = hi - lo, a.batch
    else:
        B, total = a.batch, a.batch * world
    k = n_bits.bit_length() - 1
    gen_threads = host_threads
    inputs = proofgen.random_inputs(4 + 1000 * rank, n_bits, B)  # seed 4
    t0 = time.perf_counter()
    sc, pts = proofgen.make_proofs(eng, _lib.enc_point(h), n_bits, B, inputs, threads=gen_threads)
    gen_s = time.perf_counter() - t0
    expected = corrupt(sc, pts, k, np.random.default_rng(40 + rank))
    job = eng.vjob(n_bits, 1, B, a.chunks, host_threads)
    job.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))

    def check(v):
        got = np.frombuffer(v, dtype=np.uint8)
        if not (got == expected).all():
            raise SystemExit("rank %d: verdicts differ from the expected ones at %d proofs"
                             % (rank, int((got != expected).sum())))
        return got

    for _ in range(a.warmup):
        check(job.run())
    eng.profile(True)
    barrier()
    t0 = time.perf_counter()
    results = [job.run() for _ in range(a.steps)]
    barrier()
    dt = time.perf_counter() - t0
    fixed_ms, fixed_calls, fixed_madds = eng.profile_read(0)
    eng.profile(False)
    for v in results:
        got = check(v)
    # The chunks of the timed pipeline overlap on purpose, so the event-timed duration of one
    # k_fixed_multiexp launch above includes time shared with other chunks' kernels.  One extra untimed
    # pass with a single chunk gives the same kernel's duration with the GPU to itself.
    iso = None
    api = None
    if rank == 0:
        job1 = eng.vjob(n_bits, 1, B, 1, host_threads)
        job1.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))
        check(job1.run())
        eng.profile(True)
        check(job1.run())
        iso = eng.profile_read(0)
        eng.profile(False)
        job1.close()
    if rank == 0 and world == 1 and a.extras:
        # API-inclusive rates: (1) load + run from PAGEABLE host arrays (the proofs cross PCIe and are copied into
        # pinned chunk buffers inside the timed region); (2) bp_rangeproof_verify_batch, what the Haskell shim
        # calls: cold (creates the job: pinned staging, workspaces, hashing threads) and again (job cached on the context)
        scp, ptp = sc.copy(), pts.copy()
        t0 = time.perf_counter()
        for _ in range(3):
            job.load(scp.ctypes.data_as(ctypes.c_char_p), ptp.ctypes.data_as(ctypes.c_char_p))
            check(job.run())
        dl = (time.perf_counter() - t0) / 3
        scb, ptb = sc.tobytes(), pts.tobytes()
        t0 = time.perf_counter()
        v1 = eng.verify_batch(n_bits, 1, B, scb, ptb)
        cold = time.perf_counter() - t0
        t0 = time.perf_counter()
        v2 = eng.verify_batch(n_bits, 1, B, scb, ptb)
        warm = time.perf_counter() - t0
        if not (np.array(v1, dtype=np.uint8) == expected).all() or v1 != v2:
            raise SystemExit("bp_rangeproof_verify_batch: verdicts differ from the expected ones")
        api = dict(value_incl_load=B / dl, incl_load_ms=dl * 1e3,
                   verify_batch_api=dict(cold_ms=cold * 1e3, cold_verifies_per_s=B / cold, warm_ms=warm * 1e3,
                                         warm_verifies_per_s=B / warm,
                                         what="bp_rangeproof_verify_batch from pageable host buffers: first call "
                                              "(creates and caches the job on the context), then a second call"))
    # Two jobs in flight (bp_vjob_submit / bp_vjob_wait): what a serving loop over a stream of 65,536-proof batches gets.
    # The headline above stays the synchronous step; this rate is reported beside it.  The hardware runs four of the
    # library's streams side by side, so with the host transcript two jobs in flight take TWO chunks each (four kernels
    # of twice the width): 21.8 ms per pass against 24.1 ms with four chunks each (benchmarks/pipe_sweep.sh).
    def two_in_flight(chunks):
        pj = [eng.vjob(n_bits, 1, B, chunks, host_threads) for _ in range(2)]
        for j in pj:
            j.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))
            check(j.run())
        npass = max(4, a.steps)
        barrier()
        t0 = time.perf_counter()
        pj[0].submit()
        pj[1].submit()
        outs = []
        for i in range(npass - 2):          # pass i is collected, pass i + 2 submitted on the same job
            outs.append(pj[i % 2].wait())
            pj[i % 2].submit()
        outs.append(pj[(npass - 2) % 2].wait())
        outs.append(pj[(npass - 1) % 2].wait())
        barrier()
        dp = time.perf_counter() - t0
        for v in outs:
            check(v)
        for j in pj:
            j.close()
        if dist is not None:
            tp = torch.tensor([dp], dtype=torch.float64, device="cuda")
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            dp = float(tp.item())
        return npass, dp

    pipelined = None
    other = None
    if a.extras:
        pc = a.pipe_chunks or (1 if fs_device else 2)
        npass, dp = two_in_flight(pc)
        pipelined = dict(value=total * npass / dp, unit="verifies/s", passes=npass, ms_per_pass=dp / npass * 1e3,
                         chunks_per_job=pc, transcript=a.transcript,
                         what="two jobs of %d proofs per GPU in flight on one context (bp_vjob_submit / bp_vjob_wait): "
                              "the latency-bound rounds of one pass run beside the wide kernels of the other; every "
                              "verdict checked" % B)
        # The same workload with the transcript formed at the other place, reported beside the headline.
        os.environ["BP_FS_DEVICE"] = "0" if fs_device else "1"
        oj = eng.vjob(n_bits, 1, B, 0, host_threads)
        oj.load(sc.ctypes.data_as(ctypes.c_char_p), pts.ctypes.data_as(ctypes.c_char_p))
        for _ in range(max(1, a.warmup)):
            check(oj.run())
        eng.profile(True)
        barrier()
        t0 = time.perf_counter()
        outs = [oj.run() for _ in range(a.steps)]
        barrier()
        do = time.perf_counter() - t0
        o_ms, o_calls, o_madds = eng.profile_read(0)
        eng.profile(False)
        for v in outs:
            check(v)
        oj.close()
        if dist is not None:
            to = torch.tensor([do], dtype=torch.float64, device="cuda")
            dist.all_reduce(to, op=dist.ReduceOp.MAX)
            do = float(to.item())
        opc = 2 if fs_device else 1
        onp, odp = two_in_flight(opc)
        os.environ["BP_FS_DEVICE"] = "1" if fs_device else "0"
        other = dict(transcript="host" if fs_device else "device", value=total * a.steps / do, unit="verifies/s",
                     steps=a.steps, ms_per_step=do / a.steps * 1e3,
                     two_jobs_in_flight=dict(value=total * onp / odp, ms_per_pass=odp / onp * 1e3, chunks_per_job=opc),
                     roofline=(dict(kernel="bp::k_fixed_multiexp", launches=o_calls, avg_launch_ms=o_ms / o_calls,
                                    achieved=o_madds * MODMUL_PER_MADD * MADS_PER_MODMUL / (o_ms * 1e-3) / 1e12,
                                    peak=PEAK_TMAD, unit="Tmad/s",
                                    frac=o_madds * MODMUL_PER_MADD * MADS_PER_MODMUL / (o_ms * 1e-3) / 1e12 / PEAK_TMAD,
                                    what="the same accounting as `roofline` above, over this form's timed steps")
                               if o_calls else None),
                     what=("the same %d proofs per GPU and the same checks with the Fiat-Shamir transcript formed on the "
                           "GPU (BP_FS_DEVICE=1: decimal text, SHA-256 and the challenges in HIP kernels, fsdev.h; a pass "
                           "is one in-order kernel sequence with no host step, one chunk).  north_star keeps transcript "
                           "hashing on the host and SURVEY.md 8f-1 allows the device form only as an explicit choice, so "
                           "this rate is reported beside the headline, not as it; bit-exact challenges "
                           "(tests/test_gpu_transcript_device.py)" % B) if not fs_device else
                          "the same workload with the transcript hashed by the host's threads (the library's default)")
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    job.close()

    # BASELINE config 4 as literally stated - the job's 65,536 proofs SPLIT over the GPUs - next to a weak headline
    strong = None
    if world > 1 and scaling == "weak":
        # each rank takes its share from the front of its own batch (bpdist.strong_split_slice; the 2-rank gloo test
        # checks the rule that a proof carrying its neighbour's V stays wrong inside a slice)
        scs, pts_s, exp_s = bpdist.strong_split_slice(sc, pts, expected, a.batch, rank, world)
        Bs = len(exp_s)
        js = eng.vjob(n_bits, 1, Bs, a.chunks, host_threads)
        js.load(scs.ctypes.data_as(ctypes.c_char_p), pts_s.ctypes.data_as(ctypes.c_char_p))
        first = np.frombuffer(js.run(), dtype=np.uint8)
        if not (first == exp_s).all():
            raise SystemExit("rank %d: strong-split verdicts differ from the expected ones" % rank)
        barrier()
        t0 = time.perf_counter()
        for _ in range(max(3, a.steps // 2)):
            js.run()
        barrier()
        ds = (time.perf_counter() - t0) / max(3, a.steps // 2)
        tm = torch.tensor([ds], dtype=torch.float64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ds = float(tm.item())
        js.close()
        strong = dict(scaling="strong", proofs_total=a.batch, proofs_per_gpu=Bs, ms_per_step=ds * 1e3,
                      value=a.batch / ds, unit="verifies/s",
                      what="BASELINE configs[3] literally: the 65,536 proofs split over the GPUs; small per-GPU batches "
                           "are bound by the serial challenge chain of a proof (6 rounds), not by throughput")

    msm = None
    if a.msm_logn > 0:
        logns = list(range(min(a.msm_min_logn, a.msm_logn), a.msm_logn + 1))
        msm = msm_bench(eng, rank, world, dist, logns, 5, cpu=a.cpu_sample > 0)

    if rank == 0:
        value = total * a.steps / dt
        hwq, pool, conc = eng.queue_info()
        line = {
            "metric": "range-proof verifies/sec (64-bit, batched)",
            "value": value,
            "unit": "verifies/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[3]: batch verification of %d independent %d-bit range proofs "
                            "(m=1)%s, 1%% corrupted, per-proof verdicts checked; the Fiat-Shamir transcript (%s) "
                            "inside the timed region; proofs resident in HBM + pinned host memory when the clock starts"
                            % (total, n_bits, " split over %d GPUs" % world if scaling == "strong" and world > 1 else
                               (" per GPU x %d GPUs" % world if world > 1 else ""),
                               "text, SHA-256 and challenges formed on the GPU" if fs_device else
                               "hashed by the host's threads between the kernels"),
                "transcript": "device" if fs_device else "host",
                "proofs_total": total, "proofs_per_gpu": B, "bits": n_bits, "m": 1,
                "host_threads": host_threads, "usable_cores": bpdist.usable_cores(), "ranks_on_host": local_world,
                "pinned_cores": len(pinned) if pinned else None,
                "chunks": a.chunks or (1 if fs_device else (4 if B >= 16384 else (2 if B >= 4096 else 1))),
                "proofgen_s": gen_s,
                "hw_queues_env": hwq, "stream_pool": pool, "concurrent_streams": conc,
            },
        }
        if fixed_calls:
            mads = fixed_madds * MODMUL_PER_MADD * MADS_PER_MODMUL
            ach = mads / (fixed_ms * 1e-3) / 1e12
            line["roofline"] = {
                "bound": "valu_int (32x32->64 multiply-add issue; SURVEY.md 8d: neither hbm nor mfma bounds this path)",
                "kernel": "bp::k_fixed_multiexp",
                "achieved": ach, "peak": PEAK_TMAD, "unit": "Tmad/s", "frac": ach / PEAK_TMAD,
                "launches": fixed_calls, "avg_launch_ms": fixed_ms / fixed_calls,
                "algorithmic": "mixed adds x 10 modmul x 136 mad (8-limb Montgomery accounting); "
                               "the kernel itself executes 985 v_mad_u64_u32 (and ~650 other VALU instructions) per "
                               "mixed addition on nine 29-bit limbs with the pseudo-Mersenne fold",
                "executed_mad_frac_of_peak": ach * 985.0 / 1360.0 / PEAK_TMAD,
                "traffic": None,
            }
            if iso and iso[1]:
                iso_ach = iso[2] * MODMUL_PER_MADD * MADS_PER_MODMUL / (iso[0] * 1e-3) / 1e12
                line["roofline"]["isolated"] = {
                    "what": "same kernel, one untimed single-chunk pass after the timed region (no other kernel "
                            "shares the GPU)",
                    "launches": iso[1], "avg_launch_ms": iso[0] / iso[1], "achieved": iso_ach,
                    "frac": iso_ach / PEAK_TMAD}
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):  # HBM bytes per launch from the separate rocprofv3 --pmc passes (not live)
                t = json.load(open(tf))
                line["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = t["source"]
                if "valu_issue_util" in t:  # SQ_INSTS_VALU x 4 cycles against SIMD-cycles of the same launches
                    line["roofline"]["valu_issue_util"] = t["valu_issue_util"]
        if world == 1 and a.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(sc, pts, n_bits, min(a.cpu_sample, B), got)
        if msm:
            top = msm[-1]
            line["msm"] = {
                "metric": "MSM points/sec (secp256k1, BASELINE.json configs[2])",
                "n": top["n"], "points_per_s": top["uniform"]["points_per_s"], "ms": top["uniform"]["ms"],
                "frac": top["uniform"]["frac"],
                "cpu_baseline": top.get("cpu_baseline"),
                "structured": {k2: dict(ms=top[k2]["ms"], points_per_s=top[k2]["points_per_s"]) for k2 in ("64bit", "01")},
                "resident_points_per_s": top["uniform"].get("resident_points_per_s"),
                "prepared_points_per_s": top["uniform"].get("prepared_points_per_s"),
                "prepared_frac": top["uniform"].get("prepared_frac"),
                "prepared_resident_points_per_s": top["uniform"].get("prepared_resident_points_per_s"),
                "prepared_resident_frac": top["uniform"].get("prepared_resident_frac"),
                "what": "sum e_i P_i over uniform 256-bit scalars, points resident in HBM (one-time upload excluded), "
                        "%d GPU(s) by contiguous point ranges + all-gather of 64-byte partial sums.  points_per_s / "
                        "prepared_points_per_s: scalars uploaded from pinned host memory every iteration (included); "
                        "resident_* : scalars resident as well.  prepared_*: bp_msm_points handle over the same points "
                        "(one-time shifted copies 2^(16 w) P, no window doublings at run time; N = 1 only).  "
                        "frac = SURVEY 8d algorithmic mads / (time x %d x %.1f Tmad/s)" % (world, world, PEAK_TMAD),
                "sweep": msm,
            }
        if pipelined:
            line["pipelined"] = pipelined
        if other:
            line["device_transcript" if not fs_device else "host_transcript"] = other
        if strong:
            line["strong_split"] = strong
        if api:
            line["api_inclusive"] = api
        if world == 1 and a.extras:
            try:
                line["extra"] = {"single_proof_configs": single_proof_configs(8)}
            except Exception as e:  # never lose the headline line to a side measurement
                line["extra"] = {"single_proof_configs_error": repr(e)}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
