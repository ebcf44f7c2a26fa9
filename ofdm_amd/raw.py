"""Variable-size OFDM modulator: host-side mirror of rawofdm's `raw.ofdm_params` / `raw.ofdm_mod`
(rawofdm/src/python/raw_ofdm_params.py, raw_ofdm_mod.py of the reference) on top of
ofdmtx_raw_* (include/ofdmtx.h).  The set-up arithmetic (carrier map, preamble symbols) is host
code as in the reference; mapping, preamble insertion, the shifted N-point inverse FFT, cyclic
prefix and scaling run in one HIP kernel.
"""
import cmath
import ctypes as C
import math
import random

import numpy as np
import torch

from . import _lib
from .batch import TxContext, _ptr, _stream


class ofdm_params(object):
    """rawofdm/src/python/raw_ofdm_params.py:13-99: same attributes, same values.

    options needs fft_length, occupied_tones, cp_length, num_preambles, half_sync (and may have
    verbose)."""

    def pad_symbol(self, symbol):
        padded = self.fft_length * [0, ]
        padded[self.pad: self.pad + self.occupied_tones + 1] = symbol
        return padded

    def random_symbol(self):
        # unit-modulus random phases, null at DC (raw_ofdm_params.py:29-37)
        symbol = [cmath.exp(1j * 2 * math.pi * random.random()) for _ in range(self.occupied_tones)]
        half = self.occupied_tones // 2
        return symbol[:half] + [0] + symbol[half:]

    def make_even(self, symbol):
        # only even absolute bins, boosted to keep the symbol power (raw_ofdm_params.py:39-48)
        sf = math.sqrt(2)
        for i in range(len(symbol)):
            if (self.pad + i) & 1:
                symbol[i] = 0
            else:
                symbol[i] *= sf

    def __init__(self, options):
        self.fft_length = int(math.ceil(options.fft_length / 2.0) * 2)
        self.occupied_tones = int(math.ceil(options.occupied_tones / 2.0) * 2)
        self.cp_length = options.cp_length
        self.num_preambles = options.num_preambles
        self.pad = (self.fft_length - self.occupied_tones) // 2

        random.seed(9817)                                           # raw_ofdm_params.py:61
        preambles = [self.random_symbol() for _ in range(options.num_preambles + 1)]
        self.half_sync = options.half_sync
        if self.half_sync:
            self.make_even(preambles[0])
        self.preambles = preambles
        if not self.half_sync:
            preambles = [preambles[0]] + preambles
        self.padded_preambles = [self.pad_symbol(p) for p in preambles]

        half = self.occupied_tones // 2
        carriers = half * [1, ]
        num_pilots = 0
        for i in range(5, half, 13):                                # raw_ofdm_params.py:81-85
            carriers[i] = 2
            num_pilots += 1
        carriers.extend([0] + list(reversed(carriers)))
        self.carriers = carriers
        padded_carriers = self.fft_length * [0, ]
        padded_carriers[self.pad: self.pad + self.occupied_tones + 1] = carriers
        self.padded_carriers = padded_carriers
        self.pilot_tones = num_pilots * 2
        self.data_tones = self.occupied_tones - self.pilot_tones
        self.occupied_tones += 1                                    # "include DC" (raw_ofdm_params.py:98-99)
        if getattr(options, "verbose", False):
            self._print_verbage()

    def _print_verbage(self):
        from sys import stderr
        print("\nRaw OFDM parameters:", file=stderr)
        print("FFT length:      %3d" % self.fft_length, file=stderr)
        print("Occupied Tones:  %3d (with DC)" % self.occupied_tones, file=stderr)
        print("Pilot Tones:     %3d" % self.pilot_tones, file=stderr)
        print("Data Tones:      %3d" % self.data_tones, file=stderr)
        print("CP length:       %3d" % self.cp_length, file=stderr)
        print("Extra Preambles: %3d" % self.num_preambles, file=stderr)

    @staticmethod
    def add_options(normal, expert):
        """Same option names and defaults as raw_ofdm_params.py:104-119 (optparse groups)."""
        expert.add_option("", "--fft-length", type="int", default=64,
                          help="set the number of FFT bins [default=%default]")
        expert.add_option("", "--occupied-tones", type="int", default=52,
                          help="set the number of occupied FFT bins [default=%default]")
        expert.add_option("", "--cp-length", type="int", default=16,
                          help="set the number of bits in the cyclic prefix [default=%default]")
        expert.add_option("", "--num-preambles", type="int", default=2,
                          help="number of extra preamble symbols [default=%default]")
        expert.add_option("", "--half-sync", action="store_true", default=True,
                          help="use two-halves symbol for sync [default]")
        expert.add_option("", "--full-sync", action="store_false", dest="half_sync", default=True,
                          help="use full symbol for sync")


class RawPlan(object):
    """A batch of data symbols for one carrier configuration (include/ofdmtx.h, ofdmtx_raw_*)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        s = _lib.RawSizes()
        _lib.check(self._lib.ofdmtx_raw_plan_get_sizes(self._h, C.byref(s)))
        for k, _ in _lib.RawSizes._fields_:
            setattr(self, k, int(getattr(s, k)))
        self.row_len = self.fft_length + self.cp_length

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ofdmtx_raw_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def modulate(self, data, want_mapped=False, out=None):
        """data: complex64 tensor with n_sym * data_tones elements on the plan's device.
        Returns the IQ stream (n_rows * (fft_length + cp_length) complex64)."""
        dev = self.ctx.device
        if data.dtype != torch.complex64 or data.device != dev or not data.is_contiguous():
            raise ValueError("data must be a contiguous complex64 tensor on %s" % dev)
        if data.numel() < self.n_sym * self.data_tones:
            raise ValueError("data holds %d values, plan reads %d" % (data.numel(), self.n_sym * self.data_tones))
        iq = out if out is not None else torch.empty(self.n_samples, dtype=torch.complex64, device=dev)
        if iq.numel() < self.n_samples or iq.dtype != torch.complex64:
            raise ValueError("iq buffer must be complex64 with >= %d samples" % self.n_samples)
        mapped = None
        if want_mapped:
            mapped = torch.empty((self.n_sym, self.fft_length), dtype=torch.complex64, device=dev)
        _lib.check(self._lib.ofdmtx_raw_modulate(self.ctx._h, self._h, _ptr(data), _ptr(mapped), _ptr(iq),
                                                 _stream(dev)))
        return (iq, mapped) if want_mapped else iq


def _config_arrays(carrier_map, preambles, fft_length):
    cm = np.ascontiguousarray(carrier_map, dtype=np.int32)
    if cm.shape != (fft_length,):
        raise ValueError("carrier_map must have fft_length entries")
    pre = np.ascontiguousarray(np.asarray(preambles, dtype=np.complex64).reshape(-1, fft_length))
    return cm, pre


def plan_frames(ctx, fft_length, cp_length, carrier_map, preambles, n_frames, frame_size):
    cm, pre = _config_arrays(carrier_map, preambles, fft_length)
    h = C.c_void_p()
    _lib.check(ctx._lib.ofdmtx_raw_plan_frames(ctx._h, int(fft_length), int(cp_length), cm.ctypes.data,
                                               pre.ctypes.data, pre.shape[0], int(n_frames), int(frame_size),
                                               C.byref(h)))
    return RawPlan(ctx, h)


def plan_flags(ctx, fft_length, cp_length, carrier_map, preambles, flags):
    cm, pre = _config_arrays(carrier_map, preambles, fft_length)
    fl = np.ascontiguousarray(flags, dtype=np.uint8)
    h = C.c_void_p()
    _lib.check(ctx._lib.ofdmtx_raw_plan_flags(ctx._h, int(fft_length), int(cp_length), cm.ctypes.data,
                                              pre.ctypes.data, pre.shape[0], fl.ctypes.data, len(fl), C.byref(h)))
    return RawPlan(ctx, h)


class ofdm_mod(object):
    """rawofdm/src/python/raw_ofdm_mod.py:20-52.  input[0]: data samples (data_tones per symbol),
    input[1]: start-of-frame indicator (one per symbol); output: baseband-modulated signal.

    Instead of a flowgraph the block is called: `modulate(data, flags)` for an arbitrary flag
    stream, `modulate_frames(data, frame_size)` for back-to-back frames of equal size
    (rawofdm/examples/ofdm_rxtx.py:81-87)."""

    def __init__(self, options, ctx=None, device=0):
        self.params = ofdm_params(options)
        self.ctx = ctx if ctx is not None else TxContext(device)
        self._log = bool(getattr(options, "log", False))
        self.fused = False          # whether the last modulate_bytes ran as one kernel
        self._frame_plans = {}      # (n_frames, frame_size) -> RawPlan; a plan uploads tables and transforms the preambles

    def _finish(self, plan, data):
        if self._log:                                               # raw_ofdm_mod.py:54-57 ("tx-map.dat")
            iq, mapped = plan.modulate(data, want_mapped=True)
            mapped.cpu().numpy().tofile("tx-map.dat")
            return iq
        return plan.modulate(data)

    def _plan_frames(self, n_frames, frame_size):
        """Plans for back-to-back frames are kept: a stream of equal batches builds its plan once."""
        key = (int(n_frames), int(frame_size))
        plan = self._frame_plans.get(key)
        if plan is None:
            p = self.params
            if len(self._frame_plans) >= 8:
                self._frame_plans.pop(next(iter(self._frame_plans)))
            plan = plan_frames(self.ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, *key)
            self._frame_plans[key] = plan
        return plan

    def _as_device(self, data):
        if not torch.is_tensor(data):
            data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.complex64))
        return data.to(self.ctx.device).contiguous().reshape(-1)

    def modulate(self, data, flags):
        p = self.params
        data = self._as_device(data)
        plan = plan_flags(self.ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, flags)
        return self._finish(plan, data)

    def modulate_frames(self, data, frame_size):
        p = self.params
        data = self._as_device(data)
        n_sym = data.numel() // p.data_tones
        if n_sym % frame_size:
            raise ValueError("data does not hold whole frames")
        return self._finish(self._plan_frames(n_sym // frame_size, frame_size), data)

    def modulate_bytes(self, bitrate, data, nsymbols, out=None):
        """qam_tx(bitrate, data_tones, nsymbols) feeding modulate_frames (rawofdm/examples/
        ofdm_rxtx.py:63-87), in one kernel where the geometry allows (the carrier values then
        never pass through memory), else as the two calls.  data: whole frames of
        qam_tx(...).framebytes bytes."""
        p = self.params
        tx = qam_tx(bitrate, p.data_tones, nsymbols, ctx=self.ctx)
        dev = self.ctx.device
        if not torch.is_tensor(data):
            data = torch.frombuffer(bytearray(bytes(data)) or bytearray(1), dtype=torch.uint8)
        data = data.to(dev).contiguous().reshape(-1)
        if data.dtype != torch.uint8 or data.numel() % tx.framebytes:
            raise ValueError("data must be uint8 and hold whole frames of %d bytes" % tx.framebytes)
        n_frames = data.numel() // tx.framebytes
        plan = self._plan_frames(n_frames, nsymbols)
        iq = out if out is not None else torch.empty(plan.n_samples, dtype=torch.complex64, device=dev)
        if iq.numel() < plan.n_samples or iq.dtype != torch.complex64:
            raise ValueError("iq buffer must be complex64 with >= %d samples" % plan.n_samples)
        rc = _lib.ESHAPE
        self.fused = False
        if not self._log:
            rc = self.ctx._lib.ofdmtx_raw_tx(self.ctx._h, plan._h, tx.qambits, tx.nc, tx.np, _ptr(data), _ptr(iq),
                                             _stream(dev))
        if rc == _lib.ESHAPE:
            return self._finish(plan, tx.modulate(data)) if out is None else plan.modulate(tx.modulate(data), out=out)
        _lib.check(rc)
        self.fused = True
        return iq

    add_options = staticmethod(ofdm_params.add_options)


# rawofdm/src/python/raw_qam.py:19-28: bitrate -> (bits per carrier, (nc, np))
bitrates = {
    1: (1, (4, 4)),
    2: (1, (6, 4)),   # marked "DONT USE" in the reference
    3: (2, (4, 4)),
    4: (2, (6, 4)),
    5: (4, (4, 4)),
    6: (4, (6, 4)),
    7: (6, (4, 3)),
    8: (6, (6, 4)),
}


def framebytes(bitrate, size):
    """rawofdm/src/python/raw_qam.py:30-37: bytes per frame of `size` complex values, padding
    byte included (the reference asserts divisibility; so does this)."""
    (qambits, (nc, np_)) = bitrate
    out = C.c_uint32()
    rc = _lib.load().ofdmtx_raw_framebytes(int(qambits), int(nc), int(np_), int(size), C.byref(out))
    if rc == _lib.ESHAPE:
        raise AssertionError("frame size does not give a whole number of bytes")
    _lib.check(rc)
    return out.value + 1


class qam_tx(object):
    """rawofdm/src/python/raw_qam.py:76-112: whole frames of `framebytes` bytes in, nsymbols *
    ncarriers complex values per frame out (pad -> conv_enc -> conv_punc -> intrlv_bit -> qam_enc),
    as one HIP kernel."""

    def __init__(self, bitrate, ncarriers, nsymbols, log=False, ctx=None, device=0):
        (self.qambits, (self.nc, self.np)) = bitrates[bitrate]
        self.ncarriers, self.nsymbols = int(ncarriers), int(nsymbols)
        self.framebytes = framebytes((self.qambits, (self.nc, self.np)), self.ncarriers * self.nsymbols)
        self.framebytes -= 1                                     # room for padding (raw_qam.py:83)
        self.ctx = ctx if ctx is not None else TxContext(device)
        if log:
            raise NotImplementedError("the per-block taps of qam_tx are not materialised (one fused kernel)")

    def modulate(self, data, out=None):
        """data: uint8 tensor / bytes holding whole frames.  Returns a complex64 tensor."""
        dev = self.ctx.device
        if not torch.is_tensor(data):
            data = torch.frombuffer(bytearray(bytes(data)) or bytearray(1), dtype=torch.uint8)
        data = data.to(dev).contiguous().reshape(-1)
        if data.dtype != torch.uint8 or data.numel() % self.framebytes:
            raise ValueError("data must be uint8 and hold whole frames of %d bytes" % self.framebytes)
        n_frames = data.numel() // self.framebytes
        n_out = n_frames * self.nsymbols * self.ncarriers
        vals = out if out is not None else torch.empty(n_out, dtype=torch.complex64, device=dev)
        if vals.numel() < n_out or vals.dtype != torch.complex64:
            raise ValueError("output must be complex64 with >= %d values" % n_out)
        _lib.check(self.ctx._lib.ofdmtx_raw_qam_tx(self.ctx._h, self.qambits, self.nc, self.np, self.ncarriers,
                                                   self.nsymbols, _ptr(data), n_frames, _ptr(vals), _stream(dev)))
        return vals
