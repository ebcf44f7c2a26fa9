#!/usr/bin/env python3
"""Library variants of ofdmtx_encode_bits against each other in one process (uniform batch, one regime).
usage: [AB_REGIME=8] ab_encode.py lib1.so lib2.so ..."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ofdm_amd import _lib  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402

REGIME = int(os.environ.get("AB_REGIME", "8"))
n = int(os.environ.get("AB_PACKETS", "1000000"))
ctx = TxContext(0)
plan = ctx.plan_uniform(n, REGIME, 1500)
pay = torch.randint(0, 256, (n * 1500,), dtype=torch.uint8, device=ctx.device)
msg = plan.alloc_msg()
ref = plan.alloc_msg()
plan.encode_bits(pay, out=ref)
torch.cuda.synchronize()
VP = C.c_void_p
st = VP(torch.cuda.current_stream(ctx.device).cuda_stream)
variants = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    h, p = VP(), VP()
    assert lib.ofdmtx_create(0, C.byref(h)) == 0
    assert lib.ofdmtx_plan_uniform(h, n, REGIME, 1500, 0, 4.0, _lib.PILOT_REF_STREAM, 0, _lib.F_PREAMBLE | _lib.F_ZEROGAP, 0.3, C.byref(p)) == 0
    variants.append((os.path.basename(path), lib, h, p))


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


for rep in range(3):
    for name, lib, h, p in variants:
        msg.zero_()
        t = timed(lambda: lib.ofdmtx_encode_bits(h, p, VP(pay.data_ptr()), None, None, None, VP(msg.data_ptr()), st))
        print("%-16s payload -> messages %.3f ms, identical to the default library's: %s" % (name, t, bool(torch.equal(msg, ref))), flush=True)
