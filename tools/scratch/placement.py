#!/usr/bin/env python3
"""Does the time of k_tx depend on WHICH 48 GB of HBM the samples go to?  Several output buffers in one
process (allocated one after the other, all kept), the headline kernel timed into each in turn, twice."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd.batch import TxContext  # noqa: E402

ctx = TxContext(0)
n = 1_000_000
plan = ctx.plan_uniform(n, 8, 1500)
pay = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
msg = plan.alloc_msg()
bufs = [plan.alloc_iq() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]


def timed(iq, steps=6):
    plan.tx(pay, out=iq, work=msg)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        plan.tx(pay, out=iq, work=msg)
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for rep in range(2):
    print("pass %d:" % rep, " ".join("buf%d@%#x %.3f ms" % (i, b.data_ptr(), timed(b)) for i, b in enumerate(bufs)), flush=True)
