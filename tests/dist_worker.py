"""Worker of tests/test_dist.py: one rank of a world_size-2 gloo job exercising
bulletproofs_amd.dist with a stand-in engine (the oracle plays the GPU: this test covers the
sharding / collective plumbing on CPU, the kernels have their own GPU tests)."""
import os
import random
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bp_oracle as o  # noqa: E402
from bulletproofs_amd import _lib, dist as bpdist  # noqa: E402


class OracleEngine:
    """Same byte-level surface as bulletproofs_amd.Engine for the calls dist.py makes.  The `_dev` calls take raw
    addresses like the real engine's; here they are addresses of CPU tensors (gloo), read and written with ctypes."""

    def msm_dev(self, p_scalars, p_points, n, p_out, window_bits=0):
        import ctypes
        sc = ctypes.string_at(p_scalars, 32 * n)
        pt = ctypes.string_at(p_points, 64 * n)
        ctypes.memmove(p_out, self.msm_raw(sc, pt, n), 64)

    def point_sum_dev(self, p_points, n, p_out):
        import ctypes
        acc = o.O
        for q in _lib.dec_points(ctypes.string_at(p_points, 64 * n), n):
            acc = o.point_add(acc, q)
        ctypes.memmove(p_out, _lib.enc_point(acc), 64)

    def sync(self):
        pass

    def msm_raw(self, scalars, points, n):
        return _lib.enc_point(o.sum_exps(_lib.dec_scalars(scalars, n), _lib.dec_points(points, n)))

    def add_raw(self, a, b, n):
        return _lib.enc_points([o.point_add(x, y) for x, y in zip(_lib.dec_points(a, n), _lib.dec_points(b, n))])

    def verify_batch(self, n_bits, m, batch, scalars, points):
        sc = np.frombuffer(scalars, dtype=np.uint8).reshape(5, batch, 32)
        k = (n_bits * m).bit_length() - 1
        pt = np.frombuffer(points, dtype=np.uint8).reshape(4 + 2 * k + m, batch, 64)
        out = []
        for p in range(batch):
            s = [_lib.dec_scalar(bytes(sc[r, p])) for r in range(5)]
            q = [_lib.dec_point(bytes(pt[r, p])) for r in range(4 + 2 * k + m)]
            pr = dict(tBlinding=s[0], mu=s[1], t=s[2], aCommit=q[0], sCommit=q[1], t1Commit=q[2], t2Commit=q[3],
                      productProof=dict(lCommits=q[4:4 + k], rCommits=q[4 + k:4 + 2 * k], l=s[3], r=s[4]))
            out.append(o.mrp_verify_proof(2**n_bits, q[4 + 2 * k:], pr))
        return out


def main():
    real = os.environ.get("BP_DIST_ENGINE") == "real"
    if real:  # -m gpu: the same worker over the real engine and RCCL (backend "nccl")
        import torch
        from bulletproofs_amd import Engine
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        eng = Engine(local)
        eng.set_generators(o.gs(2), o.hs(2), o.h())
    else:
        dist.init_process_group("gloo")
        eng = OracleEngine()
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = random.Random(5)  # same data on every rank
    # --- MSM sharded over point ranges ---
    n = 11
    ps = [o.point_mul(o.GEN, rng.randrange(1, o.R)) for _ in range(n)]
    es = [rng.randrange(o.R) for _ in range(n)]
    got = _lib.dec_point(bpdist.sharded_msm(eng, _lib.enc_scalars(es), _lib.enc_points(ps), n))
    assert got == o.sum_exps(es, ps), "sharded MSM mismatch on rank %d" % rank
    assert bpdist.shard_range(11, 0, 2) == (0, 6) and bpdist.shard_range(11, 1, 2) == (6, 11)
    # the resident form: every rank's shard stays where it is (HBM under nccl, CPU tensors under gloo), the 64-byte
    # partial sums are all-gathered into one tensor and summed by ONE engine call - with two gloo ranks this runs the
    # world > 1 branch of sharded_msm_dev, gather_and_sum_dev included
    import torch
    dev = "cuda" if real else "cpu"
    lo, hi = bpdist.shard_range(n, rank, world)
    d_s = torch.frombuffer(bytearray(_lib.enc_scalars(es[lo:hi])), dtype=torch.uint8).to(dev)
    d_p = torch.frombuffer(bytearray(_lib.enc_points(ps[lo:hi])), dtype=torch.uint8).to(dev)
    d_o = torch.zeros(64, dtype=torch.uint8, device=dev)
    bpdist.sharded_msm_dev(eng, d_s, d_p, hi - lo, d_o)
    eng.sync()
    assert _lib.dec_point(bytes(d_o.cpu().numpy())) == o.sum_exps(es, ps), "resident sharded MSM mismatch"
    if real:
        # the gather-and-sum step alone on a FAKE [4, 64] gathered tensor (this pool has one GPU per box): what four
        # ranks' partial sums would look like, incl. O and a cancelling pair, through bp_point_sum_dev
        fake = [ps[0], o.O, ps[1], o.point_neg(ps[0])]
        gathered = torch.frombuffer(bytearray(_lib.enc_points(fake)), dtype=torch.uint8).cuda().reshape(4, 64)
        eng.point_sum_dev(gathered.data_ptr(), 4, d_o.data_ptr())
        eng.sync()
        assert _lib.dec_point(bytes(d_o.cpu().numpy())) == ps[1], "bp_point_sum_dev over a gathered tensor"
        # bp_point_sum_dev alone: the fold of <> over a list, incl. O and a pair P, -P
        qs = ps[:5] + [o.O, o.point_neg(ps[0])]
        d_q = torch.frombuffer(bytearray(_lib.enc_points(qs)), dtype=torch.uint8).cuda()
        eng.point_sum_dev(d_q.data_ptr(), len(qs), d_o.data_ptr())
        eng.sync()
        exp = o.O
        for q in qs:
            exp = o.point_add(exp, q)
        assert _lib.dec_point(bytes(d_o.cpu().numpy())) == exp
    # --- proofs sharded over ranks ---
    nb, B = 2, 5
    proofs, vcs = [], []
    for i in range(B):
        v, g = rng.randrange(4), rng.randrange(1, o.R)
        pr = o.mrp_generate_proof(4, [(v, g)], *o.draw_blinders(rng, nb))
        if i == 3:
            pr["t"] = (pr["t"] + 1) % o.R
        proofs.append(pr)
        vcs.append(o.commit(v, g))
    k = 1
    sc = np.zeros((5, B, 32), dtype=np.uint8)
    pt = np.zeros((4 + 2 * k + 1, B, 64), dtype=np.uint8)
    for p, pr in enumerate(proofs):
        ipp = pr["productProof"]
        for r, s in enumerate([pr["tBlinding"], pr["mu"], pr["t"], ipp["l"], ipp["r"]]):
            sc[r, p] = np.frombuffer(_lib.enc_scalar(s), dtype=np.uint8)
        rows = [pr["aCommit"], pr["sCommit"], pr["t1Commit"], pr["t2Commit"]] + ipp["lCommits"] + ipp["rCommits"] + [vcs[p]]
        for r, q in enumerate(rows):
            pt[r, p] = np.frombuffer(_lib.enc_point(q), dtype=np.uint8)
    v = bpdist.sharded_verify(eng, nb, 1, B, sc, pt)
    assert v.tolist() == [1, 1, 1, 0, 1], "sharded verify mismatch on rank %d: %s" % (rank, v.tolist())
    # --- bench.py's strong split: the front of each rank's own batch, "wrong V = the neighbour's V" stays wrong ---
    exp = np.array([1, 1, 1, 0, 1], dtype=np.uint8)
    vrow = 4 + 2 * k
    for p in (0, 2):  # bench.corrupt's third kind: proof p carries the V of proof (p + 1) % B
        pt[vrow, p] = pt[vrow, (p + 1) % B]
        exp[p] = 0
    for split in (3, 5):  # a job of `split` proofs over the two ranks: shares (2, 1) and (3, 2)
        scs, pts_s, exp_s = bpdist.strong_split_slice(sc, pt, exp, split, rank, world)
        lo, hi = bpdist.shard_range(split, rank, world)
        assert scs.shape[1] == pts_s.shape[1] == len(exp_s) == hi - lo
        assert scs.flags["C_CONTIGUOUS"] and pts_s.flags["C_CONTIGUOUS"]
        got_s = eng.verify_batch(nb, 1, hi - lo, scs.tobytes(), pts_s.tobytes())
        assert [int(bool(x)) for x in got_s] == exp_s.tolist(), "strong-split verdicts on rank %d" % rank
    # --- per-rank host budget ---
    cores = bpdist.usable_cores()
    assert bpdist.host_thread_budget(1) == max(1, min(16, cores))
    assert bpdist.host_thread_budget(8) == max(1, min(16, cores // 8))
    assert sum(bpdist.host_thread_budget(world) for _ in range(world)) <= max(world, cores)
    dist.barrier()
    dist.destroy_process_group()
    if real:
        eng.close()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
