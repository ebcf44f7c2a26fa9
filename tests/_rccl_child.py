#!/usr/bin/env python3
"""One rank of the C-ABI gather test (ofdmtx_comm_* / ofdmtx_gather over RCCL), a fresh process:
   _rccl_child.py 1            one rank: communicator of size 1, the root's own placement
   _rccl_child.py N <idfile>   N ranks, rank r (RANK in the environment) on GPU r (GPU 0 for all with
                               SAME_GPU=1, which RCCL refuses); the id travels through a file"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ofdm_amd import dist as od  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402


def main():
    world = int(sys.argv[1])
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    ctx = TxContext(0 if os.environ.get("SAME_GPU") else rank)
    if world == 1:
        uid = od.Comm.unique_id()
    else:
        path = sys.argv[2]
        if rank == 0:
            uid = od.Comm.unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 30:
                    raise SystemExit("no id file")
                time.sleep(0.05)
            uid = open(path, "rb").read()
    comm = od.Comm(ctx, rank, world, uid)
    print("rank %d: communicator up" % rank, flush=True)
    n = [1000, 777, 0, 4096, 13, 8, 2, 50000][:world]
    mine = torch.full((n[rank],), complex(rank + 1, -(rank + 1)), dtype=torch.complex64, device=ctx.device)
    off = np.concatenate([[0], np.cumsum(n)])
    out = comm.gather(mine, off if rank == 0 else None, root=0)
    torch.cuda.synchronize()
    if rank == 0:
        got = out.cpu().numpy()
        want = np.concatenate([np.full(n[r], complex(r + 1, -(r + 1)), np.complex64) for r in range(world)])
        assert np.array_equal(got, want)
        print("RCCL_GATHER_OK world=%d" % world, flush=True)
    comm.close()


if __name__ == "__main__":
    main()
