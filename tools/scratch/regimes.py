#!/usr/bin/env python3
"""k_tx per regime: uniform batches of 1500-byte packets sized to ~40 GB of IQ, ms per launch and fraction
of 8 TB/s on payload + IQ bytes; also the message-input instantiation.  usage: regimes.py [regime ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import _lib  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    regimes = [int(x) for x in sys.argv[1:]] or list(range(1, 9))
    ctx = TxContext(0)
    ln = int(os.environ.get("PAYLOAD", "1500"))
    for regime in regimes:
        info = _lib.packet_info(regime, ln)
        npk = int(40e9 // (info.n_rows * 640))
        plan = ctx.plan_uniform(npk, regime, ln)
        pay = torch.randint(0, 256, (npk * ln,), dtype=torch.uint8, device=ctx.device)
        iq = plan.alloc_iq()
        msg = plan.alloc_msg()
        ms = timed(lambda: plan.tx(pay, out=iq, work=msg))
        plan.encode_bits(pay, out=msg)
        ms_msg = timed(lambda: plan.modulate(msg, out=iq))
        ms_enc = timed(lambda: plan.encode_bits(pay, out=msg))
        b = npk * (ln + info.n_rows * 640)
        bm = npk * (info.msg_len + info.n_rows * 640)
        print(json.dumps({"regime": regime, "packets": npk, "rows": info.n_rows, "ms": round(ms, 3),
                          "frac": round(b / (ms * 1e-3) / 8e12, 4), "msg_ms": round(ms_msg, 3),
                          "msg_frac": round(bm / (ms_msg * 1e-3) / 8e12, 4), "enc_ms": round(ms_enc, 3)}), flush=True)
        del plan, pay, iq, msg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
