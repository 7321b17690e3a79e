"""world_size-2 gloo test of the multi-GPU sharding plumbing (bulletproofs_amd/dist.py) on CPU."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_rank_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out.decode()[-2000:])
        assert b"ok" in out


@pytest.mark.gpu
def test_one_rank_nccl_real_engine():
    """The same worker with the real Engine and backend "nccl" (RCCL), world_size 1 (this pool has one GPU per box):
    dist.py's host-array and device-resident forms against the oracle, plus bp_point_sum_dev."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               BP_DIST_ENGINE="real")
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT)
    out, _ = p.communicate(timeout=300)
    assert p.returncode == 0, "worker failed:\n%s" % out.decode()[-3000:]
    assert b"rank 0 ok" in out
