"""bp_msm_dev_static: the entry point that allocates nothing (include/bpgpu.h).  Through the C ABI, its output must
equal bp_msm_dev's byte for byte on two calls in a row, the second with a smaller input (the static scratch array keeps
the first call's contents), on the direct path (window_bits = 0) and on the bucket path (forced window), whose
histograms, bucket lists and partial sums all live in the scratch array.  Sizes beyond the array are refused."""
import numpy as np
import pytest
import torch

import bp_oracle as o
from bulletproofs_amd import _lib

pytestmark = pytest.mark.gpu


def _inputs(engine, n, seed):
    rng = np.random.default_rng(seed)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x7F
    g = np.frombuffer(_lib.enc_point(o.GEN), dtype=np.uint8)
    pts = np.frombuffer(engine.point_mul_raw(ks.tobytes(), np.tile(g, n).tobytes(), n), dtype=np.uint8).copy()
    es = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)  # values >= r act modulo r, as in bp_msm_dev
    es[::7] = 0
    es[1::11, 8:] = 0  # 64-bit scalars
    return torch.from_numpy(es).cuda(), torch.from_numpy(pts).cuda()


def _both(engine, d_s, d_p, n, wb):
    a = torch.zeros(64, dtype=torch.uint8, device="cuda")
    b = torch.full((64,), 0xAA, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    engine.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, a.data_ptr(), wb)
    engine.msm_dev_static(d_s.data_ptr(), d_p.data_ptr(), n, b.data_ptr(), wb)
    engine.sync()
    return bytes(a.cpu().numpy()), bytes(b.cpu().numpy())


def test_static_equals_allocating_two_calls_in_a_row(engine):
    big_s, big_p = _inputs(engine, 50000, 11)
    small_s, small_p = _inputs(engine, 3000, 12)
    # known answer for the small one from the oracle (also pins bp_msm_dev itself here)
    ks = [int.from_bytes(bytes(r), "little") % o.R for r in small_s.cpu().numpy()]
    ps = _lib.dec_points(bytes(small_p.cpu().numpy()), 3000)
    exp_small = _lib.enc_point(o.sum_exps(ks[:40], ps[:40]))
    for wb_big, wb_small in [(0, 0), (13, 9), (8, 13)]:
        ref, got = _both(engine, big_s, big_p, 50000, wb_big)
        assert got == ref and ref != bytes(64)
        ref, got = _both(engine, small_s, small_p, 3000, wb_small)   # smaller input right after: stale scratch contents
        assert got == ref
        ref, got = _both(engine, small_s, small_p, 40, wb_small)
        assert got == ref == exp_small
    # n = 0 -> O, as sumExps [] [] (Utils.hs:80)
    ref, got = _both(engine, small_s, small_p, 0, 0)
    assert got == ref == bytes(64)


def test_static_rejects_what_does_not_fit(engine):
    d_s, d_p = _inputs(engine, 256, 13)
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.BpError):   # beyond BP_MSM_STATIC_MAX_N
        engine.msm_dev_static(d_s.data_ptr(), d_p.data_ptr(), 65537, out.data_ptr(), 0)
    big_s, big_p = _inputs(engine, 65536, 14)
    with pytest.raises(_lib.BpError):   # 16-bit windows: 9 x 32,768 buckets of 128 B exceed the 32 MiB array
        engine.msm_dev_static(big_s.data_ptr(), big_p.data_ptr(), 65536, out.data_ptr(), 16)
    ref, got = _both(engine, big_s, big_p, 65536, 13)  # the largest accepted size, bucket path
    assert got == ref
    ref, got = _both(engine, big_s, big_p, 65536, 0)
    assert got == ref


def test_static_calls_allocate_nothing():
    """The property the entry point exists for.  On a FRESH context (bp_msm_dev never called on it, so its grow-only
    workspace does not exist and a path that reserved it would have to allocate), 100 calls of growing n over the
    direct and the bucket plans must leave the device's free memory where it was: hipMemGetInfo is read after every
    call and must never drop below the figure taken after the first (tiny) call, which absorbs whatever the HIP
    runtime itself sets up on a kernel's first dispatch.  The results are still checked (against a second context's
    bp_msm_dev, computed before the measurement starts)."""
    from bulletproofs_amd import Engine
    ref_eng = Engine(0)
    eng = Engine(0)  # fresh: only bp_msm_dev_static will ever run on it
    try:
        d_s, d_p = _inputs(ref_eng, 65536, 21)
        sizes = [64 + (65536 - 64) * i // 99 for i in range(100)]
        plans = [(n, (0, 9, 13)[i % 3]) for i, n in enumerate(sizes)]
        outs = torch.zeros((len(plans) + 1, 64), dtype=torch.uint8, device="cuda")
        refs = torch.zeros((len(plans), 64), dtype=torch.uint8, device="cuda")
        for i, (n, wb) in enumerate(plans):
            ref_eng.msm_dev(d_s.data_ptr(), d_p.data_ptr(), n, refs[i].data_ptr(), wb)
        ref_eng.sync()
        torch.cuda.synchronize()
        eng.msm_dev_static(d_s.data_ptr(), d_p.data_ptr(), 32, outs[len(plans)].data_ptr(), 0)
        eng.sync()
        free_base = torch.cuda.mem_get_info()[0]
        lowest = free_base
        for i, (n, wb) in enumerate(plans):
            eng.msm_dev_static(d_s.data_ptr(), d_p.data_ptr(), n, outs[i].data_ptr(), wb)
            eng.sync()
            lowest = min(lowest, torch.cuda.mem_get_info()[0])
        assert lowest >= free_base, "free device memory fell by %d bytes during bp_msm_dev_static calls" % (free_base - lowest)
        assert torch.equal(outs[:len(plans)].cpu(), refs.cpu())
    finally:
        eng.close()
        ref_eng.close()
