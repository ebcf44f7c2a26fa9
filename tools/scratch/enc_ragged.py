"""Ragged k_encode_bits timing (A/B helper)."""
import numpy as np, torch
from ofdm_amd.batch import TxContext
ctx = TxContext(0)
rng = np.random.default_rng(3)
n = 400_000
regs = rng.integers(1, 9, n).astype(np.uint8)
lens = rng.integers(1, 2305, n).astype(np.uint32)
pay = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=ctx.device)
plan = ctx.plan_ragged(regs, lens)
msg = plan.alloc_msg()
for _ in range(3):
    plan.encode_bits(pay, out=msg)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    plan.encode_bits(pay, out=msg)
b.record(); torch.cuda.synchronize()
print("k_encode_bits ragged: %.3f ms" % (a.elapsed_time(b) / 10))
