"""Host-side cost of building plans (runs on the GPU box): ragged plans walk every packet."""
import time
import numpy as np
from ofdm_amd.batch import TxContext

ctx = TxContext(0)
rng = np.random.default_rng(3)
for n in (100_000, 1_000_000):
    regs = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(1, 2305, n).astype(np.uint32)
    t = time.perf_counter()
    plan = ctx.plan_ragged(regs, lens)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    lay = plan.layout()
    dl = time.perf_counter() - t
    print("plan_ragged %d packets: %.1f ms (%.0f ns/packet); layout(): %.1f ms; samples %d" % (n, dt * 1e3, dt / n * 1e9, dl * 1e3, plan.n_samples))
    del plan
t = time.perf_counter()
plan = ctx.plan_uniform(1_000_000, 8, 1500)
print("plan_uniform 1e6: %.3f ms" % ((time.perf_counter() - t) * 1e3))
import torch
for n in (100_000, 1_000_000):
    regs = torch.randint(1, 9, (n,), dtype=torch.uint8, device=ctx.device)
    lens = torch.randint(1, 2305, (n,), dtype=torch.int32, device=ctx.device)
    torch.cuda.synchronize()
    for _ in range(2):
        t = time.perf_counter()
        plan = ctx.plan_ragged(regs, lens)
        dt = time.perf_counter() - t
    print("plan_ragged on the device %d packets: %.2f ms" % (n, dt * 1e3))
