"""Where the loop-back verifier's time goes: ofdmtx_rx_demod and ofdmtx_rx_decode timed separately on the
mixed-rate ragged batch (4 10^5 packets) and on the headline batch (10^6 x 1500 bytes at 54 Mbit/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ofdm_amd.batch import TxContext

ctx = TxContext(0)
dev = ctx.device


def timed(fn, steps=3):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


rng = np.random.default_rng(3)
n = 400_000
regimes = rng.integers(1, 9, n).astype(np.uint8)
lens = rng.integers(1, 2305, n).astype(np.uint32)
for name, plan, nbytes in (("ragged 4e5", ctx.plan_ragged(regimes, lens), int(lens.sum())),
                           ("54 Mbit/s 1e6 x 1500 B", ctx.plan_uniform(1_000_000, 8, 1500), 1_500_000_000)):
    pay = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
    iq = plan.tx(pay)
    coded, status = plan.rx_demod(iq)
    t_demod = timed(lambda: plan.rx_demod(iq, out=coded))
    t_dec = timed(lambda: plan.rx_decode(coded, status))
    back, st = plan.rx_decode(coded, status)
    ok = bool(torch.equal(back, pay)) and int(st.abs().max()) == 0
    print("%-24s demod %.2f ms  decode %.2f ms  (%.1f Gsamples/s together)  clean: %s" % (name, t_demod, t_dec, plan.n_samples / (t_demod + t_dec) / 1e6, ok))
    del pay, iq, coded, status, back, st, plan
