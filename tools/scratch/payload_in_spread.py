"""Does it matter where the payload bytes lie?  The headline launch with the output in a buffer spread over
three parts, the payloads (a) in an ordinary allocation, (b) in the tail of the same spread buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ofdm_amd.batch import TxContext

ctx = TxContext(0)
dev = ctx.device
n = 1_000_000
plan = ctx.plan_uniform(n, 8, 1500)
out_bytes = plan.n_samples * 8
pay_bytes = n * 1500
raw, spread = ctx.alloc_spread(out_bytes + (1 << 26) + pay_bytes, parts=3)
print("spread:", spread, ctx.last_alloc)
iq = raw[:out_bytes].view(torch.complex64)
tail_at = (out_bytes + (1 << 26) - 1) // (1 << 26) * (1 << 26)
pay_spread = raw[tail_at:tail_at + pay_bytes]
pay_plain = torch.randint(0, 256, (pay_bytes,), dtype=torch.uint8, device=dev)
pay_spread.copy_(pay_plain)


def timed(pay, steps=60):
    for _ in range(5):
        plan.tx(pay, out=iq)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        plan.tx(pay, out=iq)
    ev[1].record(); ev[1].synchronize()
    return ev[0].elapsed_time(ev[1]) / steps


for rep in range(3):
    print("payloads in an ordinary allocation %.3f ms   in the spread buffer %.3f ms" % (timed(pay_plain), timed(pay_spread)), flush=True)
