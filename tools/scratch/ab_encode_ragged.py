#!/usr/bin/env python3
"""ofdmtx_encode_bits on the mixed-rate ragged batch, one library per process (OFDMTX_LIB selects it)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd.batch import TxContext  # noqa: E402

ctx = TxContext(0)
rng = np.random.default_rng(3)
n = 400_000
regimes = rng.integers(1, 9, n).astype(np.uint8)
lens = rng.integers(1, 2305, n).astype(np.uint32)
payloads = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=ctx.device)
plan = ctx.plan_ragged(regimes, lens)
msg = plan.alloc_msg()
ts = []
for rep in range(4):
    plan.encode_bits(payloads, out=msg)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        plan.encode_bits(payloads, out=msg)
        b.record()
    torch.cuda.synchronize()
    ts.append(float(np.median([a.elapsed_time(b) for a, b in ev])))
print(os.environ.get("OFDMTX_LIB", "default"), ["%.3f" % t for t in ts], "checksum", int(msg.to(torch.int64).sum().item()), flush=True)
