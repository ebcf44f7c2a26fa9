#!/usr/bin/env python3
"""Does the position of the PAYLOAD buffer matter as well?  Five 48 GB output buffers with a copy of the payloads
allocated after each; k_tx timed for every (payload copy, output buffer) pair."""
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
pay0 = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
msg = plan.alloc_msg()
bufs, pays = [], [pay0]
for i in range(5):
    bufs.append(plan.alloc_iq())
    if i < 4:
        pays.append(pay0.clone())


def timed(pay, iq, steps=3):
    plan.tx(pay, out=iq, work=msg)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        plan.tx(pay, out=iq, work=msg)
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


print("rows: payload copy (0 = allocated first, k = after output buffer k-1); columns: output buffer")
for j, p in enumerate(pays):
    print("payload %d:" % j, " ".join("%.3f" % timed(p, b) for b in bufs), flush=True)
