#!/usr/bin/env python3
"""rawofdm kernels of library variants against each other in ONE process, into the same spread output buffer:
modulator, bit chain alone, bytes -> IQ in one kernel (64-QAM r3/4), launched round robin.
usage: ab_libs_raw.py N lib1.so lib2.so ..."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import _lib, raw  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402

nfft = int(sys.argv[1])
ctx = TxContext(0)
dev = ctx.device


class O(object):
    num_preambles, half_sync, verbose, log = 2, True, False, False


o = O()
o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), nfft // 4
p = raw.ofdm_params(o)
nsym = 16
n_frames = ((1 << 20) if nfft <= 256 else (1 << 18)) // nsym
tx = raw.qam_tx(8, p.data_tones, nsym, ctx=ctx)
data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=dev)
plan0 = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, nsym)
arena, ok = ctx.alloc_spread(plan0.n_samples * 8 + n_frames * nsym * p.data_tones * 8)
iq = arena[:plan0.n_samples * 8].view(torch.complex64)
vals = arena[plan0.n_samples * 8:].view(torch.complex64)
vals_in = torch.empty(n_frames * nsym * p.data_tones, dtype=torch.complex64, device=dev)
tx.modulate(data, out=vals_in)
torch.cuda.synchronize()
cm, pre = raw._config_arrays(p.padded_carriers, p.padded_preambles, p.fft_length)
VP = C.c_void_p
st = VP(torch.cuda.current_stream(dev).cuda_stream)
variants = []
for path in sys.argv[2:]:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    h, rp = VP(), VP()
    assert lib.ofdmtx_create(0, C.byref(h)) == 0
    assert lib.ofdmtx_raw_plan_frames(h, p.fft_length, p.cp_length, cm.ctypes.data, pre.ctypes.data, pre.shape[0], n_frames, nsym, C.byref(rp)) == 0
    variants.append((os.path.basename(path), lib, h, rp))


def timed(fn, steps=7):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


print("N=%d spread: %s" % (nfft, ok))
for rep in range(4):
    for name, lib, h, rp in variants:
        t_m = timed(lambda: lib.ofdmtx_raw_modulate(h, rp, VP(vals_in.data_ptr()), None, VP(iq.data_ptr()), st))
        t_q = timed(lambda: lib.ofdmtx_raw_qam_tx(h, tx.qambits, tx.nc, tx.np, p.data_tones, nsym, VP(data.data_ptr()), n_frames, VP(vals.data_ptr()), st))
        t_f = timed(lambda: lib.ofdmtx_raw_tx(h, rp, tx.qambits, tx.nc, tx.np, VP(data.data_ptr()), VP(iq.data_ptr()), st))
        print("%-16s modulator %.3f | bit chain %.3f | bytes->IQ %.3f ms" % (name, t_m, t_q, t_f), flush=True)

# the variants' output against the first one's (bytes -> IQ and the modulator)
ref = None
for name, lib, h, rp in variants:
    iq.zero_()
    lib.ofdmtx_raw_tx(h, rp, tx.qambits, tx.nc, tx.np, VP(data.data_ptr()), VP(iq.data_ptr()), st)
    a = iq.clone()
    iq.zero_()
    lib.ofdmtx_raw_modulate(h, rp, VP(vals_in.data_ptr()), None, VP(iq.data_ptr()), st)
    b = iq.clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = (a, b)
    else:
        print("%-16s max |difference| to %s: bytes->IQ %.3g, modulator %.3g (largest sample %.3g)"
              % (name, variants[0][0], float((a - ref[0]).abs().max()), float((b - ref[1]).abs().max()), float(ref[0].abs().max())))
