#!/usr/bin/env python3
"""Experiment driver: rawofdm bytes -> IQ in one kernel against the two-kernel path (bit chain, then
modulator): bit identity, then time per launch of the fused kernel and of the modulator alone.
OFDMTX_LIB selects the library variant.  usage: raw_tx_sweep.py [N ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import raw  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402


def timed(fn, steps=7, warmup=2):
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
    sizes = [int(x) for x in sys.argv[1:]] or [256]
    ctx = TxContext(0)
    dev = ctx.device

    class O(object):
        num_preambles, half_sync, verbose, log = 2, True, False, False
    for nfft in sizes:
        for bitrate in (8, 5, 1):
            o = O()
            o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), nfft // 4
            mod = raw.ofdm_mod(o, ctx=ctx)
            p = mod.params
            nsym = 16
            n_frames = ((1 << 20) if nfft <= 256 else (1 << 18)) // nsym
            tx = raw.qam_tx(bitrate, p.data_tones, nsym, ctx=ctx)
            data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=dev)
            plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, nsym)
            want = torch.empty(plan.n_samples, dtype=torch.complex64, device=dev)
            vals = torch.empty(n_frames * nsym * p.data_tones, dtype=torch.complex64, device=dev)
            tx.modulate(data, out=vals)
            plan.modulate(vals, out=want)
            alg = data.numel() + plan.n_samples * 8
            iq = torch.empty_like(want)
            iq.zero_()
            mod.modulate_bytes(bitrate, data, nsym, out=iq)
            torch.cuda.synchronize()
            same = bool(torch.equal(iq.view(torch.float32), want.view(torch.float32)))
            ms = timed(lambda: mod.modulate_bytes(bitrate, data, nsym, out=iq))
            ms_mod = timed(lambda: plan.modulate(vals, out=want))
            ms_qam = timed(lambda: tx.modulate(data, out=vals))
            print(json.dumps({"N": nfft, "bitrate": bitrate, "fused": bool(mod.fused), "same": same, "ms": round(ms, 4),
                              "frac": round(alg / (ms * 1e-3) / 8e12, 4), "modulator_ms": round(ms_mod, 4), "qam_ms": round(ms_qam, 4),
                              "qam_frac": round((data.numel() + vals.numel() * 8) / (ms_qam * 1e-3) / 8e12, 4),
                              "modulator_frac": round((vals.numel() + plan.n_samples) * 8 / (ms_mod * 1e-3) / 8e12, 4)}), flush=True)
            if not same:
                return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
