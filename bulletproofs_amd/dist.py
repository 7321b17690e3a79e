"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* MSM (`sumExps`, Utils.hs:76-80): contiguous point ranges per rank, one local Pippenger each,
  then ONE tiny exchange - an all-gather of the 64-byte affine partial sums - and world-1 point
  additions.  Payload is world*64 bytes, so latency, not xGMI bandwidth, is what matters.
* Batch verification (`map verifyProof`): proofs are independent, each rank verifies its own
  slice; the only exchange is the gather of verdict bytes.
Nothing here touches the oracle or has a CPU fallback: `eng` is a bulletproofs_amd.Engine.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of n items for `rank` of `world` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device(group):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else \
        torch.device("cpu")


def sharded_msm(eng, scalars, points, n, group=None):
    """sum_i scalars[i]*points[i] over all ranks' data: `scalars`/`points` are the FULL encoded
    arrays (32/64 bytes per entry); every rank computes its range and all ranks return the sum."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n, rank, world)
    part = eng.msm_raw(scalars[32 * lo:32 * hi], points[64 * lo:64 * hi], hi - lo)
    dev = _device(group)
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(dev)
    parts = [torch.empty(64, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    # the fold of <> over the ranks' partial sums as ONE engine call: sumExps [1, 1, ..] partials
    ones = (1).to_bytes(32, "little") * world
    return eng.msm_raw(ones, b"".join(bytes(p.cpu().numpy()) for p in parts), world)


def sharded_msm_dev(eng, d_scalars, d_points, n_local, d_out, group=None, gathered=None):
    """The same with every rank's shard RESIDENT ON ITS GPU: `d_scalars` / `d_points` are torch uint8 CUDA tensors
    holding this rank's n_local scalars / points, `d_out` a 64-byte CUDA tensor that receives the total on every
    rank.  One local bucket MSM, one all-gather of the 64-byte partial sums over RCCL (device to device, no host
    round trip of the partials), one small kernel that adds the `world` points.  `gathered` may pass a reusable
    [world, 64] CUDA tensor."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        eng.msm_dev(d_scalars.data_ptr(), d_points.data_ptr(), n_local, d_out.data_ptr())
        return d_out
    part = torch.empty(64, dtype=torch.uint8, device=d_out.device)
    eng.msm_dev(d_scalars.data_ptr(), d_points.data_ptr(), n_local, part.data_ptr())
    eng.sync()  # the engine runs on its own stream; the collective is ordered behind it from the host
    return gather_and_sum_dev(eng, part, d_out, group=group, gathered=gathered)


def gather_and_sum_dev(eng, part, d_out, group=None, gathered=None):
    """The exchange step of a sharded MSM: all-gather of every rank's 64-byte partial sum (`part`, a tensor on the
    collective's device) into `gathered` [world, 64], then the sum of those `world` points by the engine
    (bp_point_sum_dev) into `d_out`.  The hand-off between torch's stream (the collective) and the engine's own stream
    is a host-side synchronisation on either side.  Runs unchanged over gloo with CPU tensors (tests/dist_worker.py
    drives it that way with two ranks); over nccl the tensors live in HBM and never touch the host."""
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world, 64), dtype=torch.uint8, device=part.device)
    dist.all_gather_into_tensor(gathered.view(-1), part, group=group)  # flat: the form gloo and nccl both take
    if gathered.is_cuda:
        torch.cuda.current_stream(gathered.device).synchronize()
    eng.point_sum_dev(gathered.data_ptr(), world, d_out.data_ptr())
    return d_out


def usable_cores():
    """Host cores this process may run on (the box's CPU share, not the machine's core count)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def host_thread_budget(local_world, cap=16):
    """Transcript-hashing threads for ONE rank: the usable cores divided by the ranks that share the host, at most
    `cap` (one 65,536-proof step fits 18 ms + 114 ms / threads, DESIGN.md 6: beyond 16 threads nothing is gained).
    Without the division every rank of an 8-GPU node would start 16 threads and ask the host for 128 cores."""
    return max(1, min(cap, usable_cores() // max(1, local_world)))


def pin_rank_cpus(local_rank, local_world):
    """Restrict this process (and the threads it starts later: hashing pool, HIP runtime) to its contiguous share of
    the usable cores, so that the ranks' hashing threads do not migrate over each other.  Returns the cores kept, or
    None when the platform has no affinity call or there are fewer cores than ranks."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if local_world <= 1 or len(cores) < local_world:
        return None
    lo, hi = shard_range(len(cores), local_rank, local_world)
    mine = cores[lo:hi]
    os.sched_setaffinity(0, mine)
    return mine


def strong_split_slice(scalars, points, expected, batch, rank, world):
    """The strong-scaling leg of bench.py: this rank's share of a `batch`-proof job cut from the FRONT of the rank's
    own SoA arrays (scalars[5, B, 32], points[rows, B, 64], expected[B]; B >= the share).  The slice is self-contained:
    the corruption rule "proof p carries its neighbour's V" copied the neighbour's bytes into row V of proof p when the
    batch was built, so proof p stays wrong without its neighbour.  -> (scalars, points, expected) of the share,
    C-contiguous."""
    import numpy as np
    lo, hi = shard_range(batch, rank, world)
    n = hi - lo
    if n > scalars.shape[1]:
        raise ValueError("the rank holds %d proofs, its share of the split is %d" % (scalars.shape[1], n))
    return (np.ascontiguousarray(scalars[:, :n]), np.ascontiguousarray(points[:, :n]), expected[:n].copy())


def sharded_verify(eng, n_bits, m, batch, scalars, points, group=None):
    """verdicts for `batch` proofs (numpy SoA arrays scalars[5,B,32], points[rows,B,64]); each rank
    verifies its contiguous slice and all ranks return all verdicts."""
    import numpy as np
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(batch, rank, world)
    sc = np.ascontiguousarray(scalars[:, lo:hi])
    pt = np.ascontiguousarray(points[:, lo:hi])
    mine = np.frombuffer(bytes(bytearray(int(b) for b in eng.verify_batch(n_bits, m, hi - lo, sc.tobytes(),
                                                                          pt.tobytes()))), dtype=np.uint8) \
        if hi > lo else np.zeros(0, dtype=np.uint8)
    dev = _device(group)
    width = -(-batch // world)
    buf = torch.zeros(width, dtype=torch.uint8, device=dev)
    buf[:hi - lo] = torch.from_numpy(mine.copy()).to(dev)
    parts = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r, p in enumerate(parts):
        a, b = shard_range(batch, r, world)
        out.append(p[:b - a].cpu().numpy())
    return np.concatenate(out)
