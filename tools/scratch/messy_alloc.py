"""Placement on a device with a history: the memory is taken in pieces of `piece` MiB through the
caching allocator, given back in a random order, and ofdmtx_stream_alloc has to find the parts in what the
driver then hands out.  Prints the placement record and the time of the headline launch into the buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ofdm_amd.batch import TxContext

ctx = TxContext(0)
dev = ctx.device
plan = ctx.plan_uniform(1_000_000, 8, 1500)
pay = torch.randint(0, 256, (1_500_000_000,), dtype=torch.uint8, device=dev)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
piece = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for rnd in range(2):
    free, total = torch.cuda.mem_get_info(dev)
    n = int((free - (70 << 30)) // (piece << 20)) if rnd == 0 else int((free - (8 << 30)) // (piece << 20))
    held = [torch.empty(piece << 20, dtype=torch.uint8, device=dev) for _ in range(n)]
    for i in rng.permutation(n):
        held[i] = None
        if rnd == 1 and i % 8 == 0:
            torch.cuda.empty_cache()                         # given back a few at a time, in that order
    del held
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    raw, spread = ctx.alloc_spread(plan.n_samples * 8, parts=3)
    dt = time.perf_counter() - t0
    iq = raw.view(torch.complex64)
    plan.tx(pay, out=iq)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        plan.tx(pay, out=iq)
    ev[1].record(); ev[1].synchronize()
    print("round %d: %d pieces of %d MiB shuffled; spread=%s in %.2f s, %s; k_tx %.3f ms" % (rnd, n, piece, spread, dt, ctx.last_alloc, ev[0].elapsed_time(ev[1]) / 3), flush=True)
    del iq, raw
