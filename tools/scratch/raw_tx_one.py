#!/usr/bin/env python3
"""One rawofdm bytes -> IQ configuration, a few launches (for rocprofv3 passes).
usage: raw_tx_one.py N bitrate [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import raw  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402

nfft, bitrate = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = TxContext(0)


class O(object):
    num_preambles, half_sync, verbose, log = 2, True, False, False


o = O()
o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), nfft // 4
mod = raw.ofdm_mod(o, ctx=ctx)
p = mod.params
nsym = 16
n_frames = ((1 << 20) if nfft <= 256 else (1 << 18)) // nsym
tx = raw.qam_tx(bitrate, p.data_tones, nsym, ctx=ctx)
data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=ctx.device)
plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, nsym)
iq = torch.empty(plan.n_samples, dtype=torch.complex64, device=ctx.device)
vals = torch.empty(n_frames * nsym * p.data_tones, dtype=torch.complex64, device=ctx.device)
for _ in range(steps):
    mod.modulate_bytes(bitrate, data, nsym, out=iq)
    tx.modulate(data, out=vals)
    plan.modulate(vals, out=iq)
torch.cuda.synchronize()
print("done", mod.fused, data.numel() + plan.n_samples * 8)
