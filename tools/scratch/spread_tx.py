#!/usr/bin/env python3
"""k_tx (10^6 x 1500-byte packets, 54 Mbit/s) into a spread output buffer (TxPlan.alloc_iq_spread) and into
ordinary allocations made after it; also the message-input instantiation and bit identity."""
import os
import sys
import time

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
t0 = time.perf_counter()
spread, ok = plan.alloc_iq_spread()
torch.cuda.synchronize()
print("alloc_iq_spread: %.2f s, spread=%s, ptr %#x" % (time.perf_counter() - t0, ok, spread.data_ptr()), flush=True)
two, ok2 = plan.alloc_iq_spread(thirds=3)
pay_far, ok3 = ctx.alloc_spread(pay.numel(), thirds=4)
pay_far.copy_(pay)
msg_far, _ = ctx.alloc_spread(msg.numel(), thirds=4)
print("two thirds: spread=%s; payload copy in the last third: spread=%s" % (ok2, ok3), flush=True)
plain = [plan.alloc_iq() for _ in range(2)]
for name, b, p_, m_ in [("out 2/3, payload last third", two, pay_far, msg_far), ("out 2/3, payload plain", two, pay, msg),
                        ("out 3/3, payload last third", spread, pay_far, msg_far)]:
    plan.tx(p_, out=b, work=m_)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    torch.cuda.synchronize()
    for a_, b_ in evs:
        a_.record()
        plan.tx(p_, out=b, work=m_)
        b_.record()
    torch.cuda.synchronize()
    ms = float(np.median([a_.elapsed_time(b_) for a_, b_ in evs]))
    print("%-30s payload -> IQ %.3f ms = %.1f %%" % (name, ms, 49.5e9 / ms / 8e12 * 100 * 1e3), flush=True)


def timed(fn, steps=5):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for name, b in [("spread", spread), ("plain 0", plain[0]), ("plain 1", plain[1])]:
    ms = timed(lambda: plan.tx(pay, out=b, work=msg))
    print("%-8s payload -> IQ %.3f ms = %.1f %% of 8 TB/s" % (name, ms, 49.5e9 / ms / 8e12 * 100 * 1e3), flush=True)
plan.encode_bits(pay, out=msg)
for name, b in [("spread", spread), ("plain 0", plain[0])]:
    ms = timed(lambda: plan.modulate(msg, out=b))
    print("%-8s messages -> IQ %.3f ms" % (name, ms), flush=True)
plan.tx(pay, out=spread, work=msg)
plan.tx(pay, out=plain[0], work=msg)
torch.cuda.synchronize()
print("identical:", bool(torch.equal(spread.view(torch.float32), plain[0].view(torch.float32))))
del spread
torch.cuda.synchronize()
print("free after release: %.1f GB" % (torch.cuda.mem_get_info(ctx.device)[0] / 1e9))
