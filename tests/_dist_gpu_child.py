"""One rank of tests/test_dist_gpu.py: started as a fresh process (before anything touches the GPU),
joins a gloo group with its sibling, produces its packet range with the HIP kernels in chunks
through ofdm_amd.dist.gather_chunked, and -- on rank 0 -- compares the gathered stream with ONE launch
over the whole batch.  The ranks share GPU 0 (the box has one); with OFDM_TEST_BACKEND=nccl each rank owns GPU `rank`
and the chunks travel over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ofdm_amd import _lib  # noqa: E402
from ofdm_amd import dist as od  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402


def main():
    mode, n_chunks = sys.argv[1], int(sys.argv[2])
    backend = os.environ.get("OFDM_TEST_BACKEND", "gloo")
    rank = int(os.environ["RANK"])
    dev = rank if backend == "nccl" else 0               # RCCL: one GPU per rank; gloo: the ranks share GPU 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    world = dist.get_world_size()
    ctx = TxContext(dev)
    rng = np.random.default_rng(99)                      # the same batch in every rank
    n = 203
    if mode == "uniform":
        regime, ln = 8, 1500
        regimes, lens = np.full(n, regime, np.uint8), np.full(n, ln, np.uint32)
    else:
        regimes, lens = rng.integers(1, 9, n).astype(np.uint8), rng.integers(0, 900, n).astype(np.uint32)
    payloads = torch.from_numpy(rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)).to(ctx.device)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n_sym1 = np.array([_lib.packet_info(int(r), int(l)).n_sym + 1 for r, l in zip(regimes, lens)])
    sym_before = np.concatenate([[0], np.cumsum(n_sym1)])

    lo, hi = od.shard_bounds(n, world, rank)
    spans = [(lo + a, lo + b) for a, b in od.chunk_bounds(hi - lo, n_chunks)]
    plans = []
    for a, b in spans:
        start = od.pilot_start_of_shard(int(sym_before[a]), 5)
        if mode == "uniform":
            plans.append(ctx.plan_uniform(b - a, int(regimes[0]), int(lens[0]), pilot_start=start))
        else:
            plans.append(ctx.plan_ragged(regimes[a:b], lens[a:b], pilot_start=start))

    def produce(k, buf):
        a, _ = spans[k]
        plans[k].tx(payloads[int(offs[a]):], out=buf)

    got = od.gather_chunked(produce, [p.n_samples for p in plans], dst=0, device=ctx.device)
    torch.cuda.synchronize()
    if rank == 0:
        whole = ctx.plan_ragged(regimes, lens, pilot_start=5) if mode != "uniform" else \
            ctx.plan_uniform(n, int(regimes[0]), int(lens[0]), pilot_start=5)
        want = whole.tx(payloads)
        torch.cuda.synchronize()
        assert got.numel() == want.numel(), (got.numel(), want.numel())
        assert torch.equal(got.view(torch.float32), want.view(torch.float32)), "gathered stream differs"
        print("GATHER_OK %s chunks=%d samples=%d" % (mode, n_chunks, got.numel()), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
