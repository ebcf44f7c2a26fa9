#!/usr/bin/env python3
"""Library variants against each other in ONE process, into the same buffers (a spread output buffer and an
ordinary one from the default library): every variant is loaded with ctypes, gets its own context and plan,
and ofdmtx_tx is launched round robin.  usage: ab_libs.py lib1.so lib2.so ..."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import _lib  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402

ctx = TxContext(0)
REGIME = int(os.environ.get("AB_REGIME", "8"))       # AB_REGIME / AB_PACKETS: another rate / batch size
n = int(os.environ.get("AB_PACKETS", "1000000"))
plan = ctx.plan_uniform(n, REGIME, 1500)
out, ok = plan.alloc_iq_spread()
plain = plan.alloc_iq()
pay = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
msg = plan.alloc_msg()
plan.encode_bits(pay, out=msg)
torch.cuda.synchronize()
VP = C.c_void_p
variants = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    h, p = VP(), VP()
    assert lib.ofdmtx_create(0, C.byref(h)) == 0
    assert lib.ofdmtx_plan_uniform(h, n, REGIME, 1500, 0, 4.0, _lib.PILOT_REF_STREAM, 0, _lib.F_PREAMBLE | _lib.F_ZEROGAP, 0.3, C.byref(p)) == 0
    p0 = VP()
    assert lib.ofdmtx_plan_uniform(h, n, REGIME, 1500, 0, 4.0, _lib.PILOT_REF_STREAM, 0, 0, 0.3, C.byref(p0)) == 0
    variants.append((os.path.basename(path), lib, h, p, p0))
st = VP(torch.cuda.current_stream(ctx.device).cuda_stream)


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


print("spread:", ok)
for rep in range(4):
    for name, lib, h, p, p0 in variants:
        t_s = timed(lambda: lib.ofdmtx_tx(h, p, VP(pay.data_ptr()), VP(msg.data_ptr()), VP(out.data_ptr()), st))
        t_p = timed(lambda: lib.ofdmtx_tx(h, p, VP(pay.data_ptr()), VP(msg.data_ptr()), VP(plain.data_ptr()), st))
        t_m = timed(lambda: lib.ofdmtx_modulate(h, p, VP(msg.data_ptr()), None, VP(out.data_ptr()), st))
        t_0 = timed(lambda: lib.ofdmtx_modulate(h, p0, VP(msg.data_ptr()), None, VP(out.data_ptr()), st))
        print("%-14s payload->IQ spread %.3f, ordinary %.3f; messages->IQ spread %.3f, symbol rows only %.3f ms" % (name, t_s, t_p, t_m, t_0), flush=True)
