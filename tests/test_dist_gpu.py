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


def _gpu_count():
    """GPUs the multi-GPU cases may use: they have never run on hardware (every box this suite has seen
    has one GPU), so they are opt-in -- OFDM_TEST_MULTI_GPU=1 on a node with two or more GPUs."""
    if os.environ.get("OFDM_TEST_MULTI_GPU") != "1":
        return 1
    import torch
    return torch.cuda.device_count()                     # (counting devices does not initialise the GPU)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("mode,n_chunks", [("uniform", 1), ("uniform", 4), ("ragged", 3)])
def test_two_ranks_hip_producer_chunked_gather(mode, n_chunks, backend):
    if backend == "nccl" and _gpu_count() < 2:
        pytest.skip("RCCL wants one GPU per rank (OFDM_TEST_MULTI_GPU=1 on a multi-GPU node)")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OFDM_TEST_BACKEND=backend)
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


def _run_rccl_children(world, tmp_path, same_gpu=False):
    idfile = str(tmp_path / "comm_id")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank))
        if same_gpu:
            env["SAME_GPU"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py"), str(world), idfile],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=100))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


def test_abi_gather_one_rank_brings_rccl_up(tmp_path):
    """ofdmtx_comm_unique_id / ofdmtx_comm_create / ofdmtx_gather on hardware: librccl found at run
    time, a communicator of one rank, the root's own shard placed."""
    procs, outs = _run_rccl_children(1, tmp_path)
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    assert "RCCL_GATHER_OK world=1" in outs[0][0]


def test_abi_gather_two_ranks_on_one_gpu_is_refused_cleanly(tmp_path):
    """RCCL does not take two ranks on one GPU: the ABI hands its verdict back as OFDMTX_ECOMM with the
    ncclResult (5, invalid usage) instead of hanging or aborting."""
    procs, outs = _run_rccl_children(2, tmp_path, same_gpu=True)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode != 0
        assert "code -9" in se and "ncclResult 5" in se, se[-2000:]


def test_abi_gather_ranks_on_their_own_gpus(tmp_path):
    world = min(_gpu_count(), 8)
    if world < 2:
        pytest.skip("needs two GPUs (OFDM_TEST_MULTI_GPU=1 on a multi-GPU node)")
    procs, outs = _run_rccl_children(world, tmp_path)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert "RCCL_GATHER_OK world=%d" % world in outs[0][0]
