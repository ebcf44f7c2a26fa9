"""The multi-rank path with the HIP kernels as producer: two fresh child processes (started before
any GPU call, one process group over gloo, both on GPU 0) shard a batch by packet index, produce
their ranges in chunks and gather them on rank 0 with the chunk-overlapped exchange of
ofdm_amd.dist.gather_chunked; rank 0 checks the result bit for bit against one launch over the whole
batch (pilot counter continued across shards and chunks)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,n_chunks", [("uniform", 1), ("uniform", 4), ("ragged", 3)])
def test_two_ranks_hip_producer_chunked_gather(mode, n_chunks):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_child.py"), mode,
                                       str(n_chunks)], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=100))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert "GATHER_OK" in outs[0][0]
