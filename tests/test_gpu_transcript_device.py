"""GPU parity tests of the Fiat-Shamir transcript formed on the device (fsdev.h, verify.hip k_fs_initial / k_fs_round,
vbatch_run_fsdev_async): batches above the latency path's size run a pass with no host step, and their verdicts must
equal the oracle's verifyProof and the host-transcript form's (BP_FS_DEVICE=0), proof by proof.  An honest proof only
verifies when every challenge (y, z, x, u and the x_r of each round) is the reference's, bit for bit."""
import os
import random

import pytest

import bp_oracle as o
from test_gpu_verify import pack_batch, mutate, degenerate_lr

pytestmark = pytest.mark.gpu
R = o.R


def _cases(rng, n, m):
    m2 = 2 ** o.log2_ceil(m)
    proofs, vcs = [], []
    for _ in range(3):
        vs = [(rng.randrange(0, 2**n), rng.randrange(1, R)) for _ in range(m)]
        proofs.append(o.mrp_generate_proof(2**n, vs, *o.draw_blinders(rng, n * m2)))
        vcs.append([o.commit(v, b) for v, b in vs])
    bad = [mutate(rng, proofs[i % 3], vcs[i % 3]) for i in range(5)]
    all_p = proofs + [b[0] for b in bad]
    all_v = vcs + [b[1] for b in bad]
    k = len(proofs[0]["productProof"]["lCommits"])
    for kind in range(5 if k else 0):       # L_r / R_r = O, equal, opposite: "O" in the text, exceptional additions
        all_p.append(degenerate_lr(proofs[kind % 3], kind, kind % k))
        all_v.append(vcs[kind % 3])
    return m2, all_p, all_v


@pytest.mark.parametrize("style", ["derived", "plain"])
def test_device_transcript_verdicts(engine, style):
    old_style = o.SHOW_STYLE
    old_env = os.environ.get("BP_FS_DEVICE")
    try:
        o.SHOW_STYLE = style
        o.reset_generators()
        engine.set_show_style(0 if style == "derived" else 1)
        engine.set_generators(o.gs(64), o.hs(64), o.h())
        rng = random.Random(4100 + len(style))
        for n, m in [(8, 1), (4, 2), (2, 3), (1, 1), (16, 4)]:
            m2, all_p, all_v = _cases(rng, n, m)
            exp = [o.mrp_verify_proof(2**n, v, p) for p, v in zip(all_p, all_v)]
            assert exp[:3] == [True] * 3
            copies = 800 // len(all_p) + 1          # above the latency path's 768 proofs
            sc, pts, _ = pack_batch(all_p * copies, all_v * copies, m2)
            B = len(all_p) * copies
            for mode, chunks in (("1", 0), ("1", 3), ("0", 0)):
                os.environ["BP_FS_DEVICE"] = mode
                job = engine.vjob(n, m2, B, chunks, 4)
                job.load(sc, pts)
                assert [b != 0 for b in job.run()] == exp * copies, (n, m, mode, chunks)
                assert [b != 0 for b in job.run()] == exp * copies          # re-runnable
                job.submit()                                                 # and through the driver thread
                assert [b != 0 for b in job.wait()] == exp * copies
                job.close()
    finally:
        if old_env is None:
            os.environ.pop("BP_FS_DEVICE", None)
        else:
            os.environ["BP_FS_DEVICE"] = old_env
        o.SHOW_STYLE = old_style
        o.reset_generators()
        engine.set_show_style(0 if old_style == "derived" else 1)
        engine.set_generators(o.gs(128), o.hs(128), o.h())
