#!/usr/bin/env python3
"""Does k_tx speed up while a fresh box warms up?  Launches the headline kernel back to back for a few
seconds and prints the launch time against the time since the first launch."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd.batch import TxContext  # noqa: E402

ctx = TxContext(0)
n = 1_000_000
plan = ctx.plan_uniform(n, 8, 1500)
pay = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
iq = plan.alloc_iq()
msg = plan.alloc_msg()
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 8.0:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    plan.tx(pay, out=iq, work=msg)
    b.record()
    torch.cuda.synchronize()
    if k < 10 or k % 20 == 0:
        print("t=%6.2f s  launch %4d  %.3f ms" % (time.perf_counter() - t0, k, a.elapsed_time(b)), flush=True)
    k += 1
