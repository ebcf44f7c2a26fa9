"""Host cost of one ofdmtx_tx call (small plan, back-to-back launches)."""
import time
import torch
from ofdm_amd.batch import TxContext
ctx = TxContext(0)
for kind in ("uniform", "ragged"):
    if kind == "uniform":
        plan = ctx.plan_uniform(64, 8, 100)
    else:
        import numpy as np
        plan = ctx.plan_ragged(np.full(64, 8, np.uint8), np.full(64, 100, np.uint32))
    pay = torch.zeros(plan.payload_bytes, dtype=torch.uint8, device=ctx.device)
    iq = plan.alloc_iq()
    for _ in range(100):
        plan.tx(pay, out=iq)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000):
        plan.tx(pay, out=iq)
    t1 = time.perf_counter() - t
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t
    print("%s: %.1f us per call issued, %.1f us per call completed" % (kind, t1 / 2000 * 1e6, t2 / 2000 * 1e6))
