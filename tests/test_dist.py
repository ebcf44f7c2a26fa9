"""Packet sharding across ranks: world_size-2 gloo run on CPU.  Each rank produces the IQ of
its packet range (the oracle stands in for the GPU kernels here), the shards are gathered on
rank 0 and must equal the single-stream result, pilot counter continuity included."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ofdm_amd import dist as od
from oracle import ftw_oracle as o


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            spans = [od.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    rows = np.random.default_rng(0).integers(20, 1100, 500)
    for world in (2, 8):
        spans = [od.shard_bounds_by_rows(rows, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == len(rows)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        loads = [rows[a:b].sum() for a, b in spans]
        assert max(loads) - min(loads) <= 2 * rows.max()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, payloads, regime, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = od.shard_bounds(len(payloads), world, rank)
        n_sym_before = sum(o.get_info(p, regime, 4.0)["N_sym"] + 1 for p in payloads[:lo])
        off = od.pilot_start_of_shard(n_sym_before)
        parts = []
        for p in payloads[lo:hi]:
            r = o.tx_packet(p, regime, 4.0, pilot_offset=off)
            off += r["info"]["N_sym"] + 1
            parts.append(r["iq"].astype(np.complex64))
        mine = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.complex64))
        got = od.gather_iq(mine, dst=0)
        if rank == 0:
            q.put(got.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_single_stream():
    rng = np.random.default_rng(4)
    payloads = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(0, 300, 7)]
    regime = "4"
    want = o.tx_stream(payloads, regime).astype(np.complex64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, payloads, regime, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, want)


def _raw_worker(rank, world, port, data, size, q):
    """rawofdm: frames are independent (no state runs from one frame to the next), so a batch is
    sharded by frame index exactly like packets."""
    from oracle import raw_oracle as ro
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = ro.OfdmParams(ro.Options(fft_length=64, occupied_tones=52, cp_length=16))
        n_frames = data.shape[0] // size
        lo, hi = od.shard_bounds(n_frames, world, rank)
        rows = ro.modulate(params, data[lo * size:hi * size], ro.frame_flags(hi - lo, size)) if hi > lo else np.zeros((0, 80))
        mine = torch.from_numpy(np.asarray(rows).astype(np.complex64).reshape(-1))
        got = od.gather_iq(mine, dst=0)
        if rank == 0:
            q.put(got.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_rawofdm_frames_equal_single_stream():
    from oracle import raw_oracle as ro
    params = ro.OfdmParams(ro.Options(fft_length=64, occupied_tones=52, cp_length=16))
    rng = np.random.default_rng(5)
    size, n_frames = 4, 5
    data = ((rng.integers(0, 2, (n_frames * size, params.data_tones)) * 2 - 1) +
            1j * (rng.integers(0, 2, (n_frames * size, params.data_tones)) * 2 - 1)) / np.sqrt(2)
    want = np.asarray(ro.modulate(params, data, ro.frame_flags(n_frames, size))).astype(np.complex64).reshape(-1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_raw_worker, args=(r, 2, port, data, size, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, want)


def _chunk_worker(rank, world, port, payloads, regime, n_chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = od.shard_bounds(len(payloads), world, rank)
        n_sym = [o.get_info(p, regime, 4.0)["N_sym"] for p in payloads]
        spans = [(lo + a, lo + b) for a, b in od.chunk_bounds(hi - lo, n_chunks)]
        sizes = [sum(o.n_samples(n_sym[i]) for i in range(a, b)) for a, b in spans]

        def produce(k, buf):
            a, b = spans[k]
            off = od.pilot_start_of_shard(sum(n + 1 for n in n_sym[:a]))
            parts = []
            for p in payloads[a:b]:
                r = o.tx_packet(p, regime, 4.0, pilot_offset=off)
                off += r["info"]["N_sym"] + 1
                parts.append(r["iq"].astype(np.complex64))
            buf.copy_(torch.from_numpy(np.concatenate(parts)))

        got = od.gather_chunked(produce, sizes, dst=0, device="cpu")
        if rank == 0:
            q.put(got.numpy())
    finally:
        dist.destroy_process_group()


def test_chunk_bounds():
    for n in (0, 1, 5, 1000):
        for k in (1, 3, 8, 2000):
            spans = od.chunk_bounds(n, k)
            assert spans[0][0] == 0 and spans[-1][1] == n and len(spans) <= max(n, 1)
            assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))


def test_two_rank_chunked_gather_equals_single_stream():
    """The chunk-overlapped gather (peers send chunk k while producing chunk k + 1, the root receives
    into place): same result as one stream, chunk counts that do and do not divide the shard."""
    rng = np.random.default_rng(6)
    payloads = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(0, 200, 11)]
    regime = "6"
    want = o.tx_stream(payloads, regime).astype(np.complex64)
    for n_chunks in (1, 3, 4):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, payloads, regime, n_chunks, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert np.array_equal(got, want), n_chunks
