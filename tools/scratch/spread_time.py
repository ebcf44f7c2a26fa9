#!/usr/bin/env python3
"""k_tx (10^6 x 1500-byte packets, 54 Mbit/s) into a spread output buffer: median ms of 9 launches.  OFDMTX_LIB
selects the library variant."""
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
out, ok = plan.alloc_iq_spread()
pay = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
msg = plan.alloc_msg()
plain = plan.alloc_iq()


def timed(fn, steps=9):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


t_s = timed(lambda: plan.tx(pay, out=out, work=msg))
t_p = timed(lambda: plan.tx(pay, out=plain, work=msg))
plan.encode_bits(pay, out=msg)
t_m = timed(lambda: plan.modulate(msg, out=out))
print("%-12s spread=%s  payload->IQ spread %.3f ms, ordinary %.3f ms; messages->IQ spread %.3f ms"
      % (os.path.basename(os.environ.get("OFDMTX_LIB", "libofdmtx.so")), ok, t_s, t_p, t_m), flush=True)
