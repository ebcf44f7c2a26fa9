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


def _even_up(n):
    """Smallest even integer >= n."""
    n = int(math.ceil(n))
    return n + (n & 1)


class ofdm_params(object):
    """Carrier layout and preamble symbols of the variable-size modulator: the attributes and
    values of rawofdm/src/python/raw_ofdm_params.py:13-99, derived with index arithmetic.

    Layout of the fft_length bins (DC in the middle, raw_ofdm_params.py:20-22): `pad` unused bins,
    occupied/2 carriers, DC (null), occupied/2 carriers, pad - 1 unused bins.  With m = 1 ..
    occupied/2 counting outwards from the band edge towards DC on either side, carriers with
    m - 1 = 5, 18, 31, ... are pilots (:81-85), the layout is mirrored about DC (:88).
    Preamble symbols: unit-modulus values exp(2 pi j u) on every occupied carrier, u taken in
    ascending carrier order, symbol after symbol, from ONE stream random.Random(9817) -- the stream
    random.seed(9817) gives the reference (:29-37,61-62); the first symbol keeps only even bins,
    times sqrt(2), when half_sync is set (:39-48,64-67), otherwise it is sent twice (:71-73).

    options needs fft_length, occupied_tones, cp_length, num_preambles, half_sync (and may have
    verbose)."""

    def __init__(self, options):
        n_fft = _even_up(options.fft_length)
        n_occ = _even_up(options.occupied_tones)                    # without DC
        half = n_occ // 2
        self.fft_length = n_fft
        self.cp_length = options.cp_length
        self.num_preambles = options.num_preambles
        self.half_sync = options.half_sync
        self.pad = (n_fft - n_occ) // 2
        window = slice(self.pad, self.pad + n_occ + 1)              # occupied bins, DC included
        not_dc = np.arange(n_occ + 1) != half

        # ---- preamble symbols: all phases in one draw, in the reference's order
        n_pre = int(options.num_preambles) + 1
        stream = random.Random(9817)
        phases = [stream.random() for _ in range(n_pre * n_occ)]
        table = np.zeros((n_pre, n_occ + 1), dtype=complex)
        table[:, not_dc] = np.array([cmath.exp(2j * math.pi * u) for u in phases]).reshape(n_pre, n_occ)
        if self.half_sync:
            odd_bin = ((self.pad + np.arange(n_occ + 1)) & 1).astype(bool)
            table[0, odd_bin] = 0
            table[0, ~odd_bin] *= math.sqrt(2)
        self.preambles = table.tolist()
        sent = table if self.half_sync else np.concatenate([table[:1], table])
        padded = np.zeros((len(sent), n_fft), dtype=complex)
        padded[:, window] = sent
        self.padded_preambles = padded.tolist()

        # ---- carrier map: 1 data, 2 pilot, 0 unused
        side = np.ones(half, dtype=int)
        side[5::13] = 2
        carriers = np.concatenate([side, [0], side[::-1]])
        self.carriers = carriers.tolist()
        padded_map = np.zeros(n_fft, dtype=int)
        padded_map[window] = carriers
        self.padded_carriers = padded_map.tolist()
        self.pilot_tones = int((carriers == 2).sum())
        self.data_tones = int((carriers == 1).sum())
        self.occupied_tones = n_occ + 1                             # sizes downstream count DC (:98-99)
        if getattr(options, "verbose", False):
            self._report()

    def _report(self):
        import sys
        sys.stderr.write("\nRaw OFDM parameters:\n" + "".join("%-16s %3d%s\n" % row for row in (
            ("FFT length:", self.fft_length, ""), ("Occupied Tones:", self.occupied_tones, " (with DC)"),
            ("Pilot Tones:", self.pilot_tones, ""), ("Data Tones:", self.data_tones, ""),
            ("CP length:", self.cp_length, ""), ("Extra Preambles:", self.num_preambles, ""))))

    _OPTIONS = (  # name, default, what it sets (names and defaults of raw_ofdm_params.py:104-119)
        ("--fft-length", 64, "FFT size N (bins per OFDM symbol)"),
        ("--occupied-tones", 52, "carriers in use, DC not counted"),
        ("--cp-length", 16, "cyclic prefix, in samples"),
        ("--num-preambles", 2, "preamble symbols behind the synchronisation symbol"),
    )

    @staticmethod
    def add_options(normal, expert):
        for name, default, what in ofdm_params._OPTIONS:
            expert.add_option("", name, type="int", default=default, help=what + " [default=%default]")
        expert.add_option("", "--half-sync", action="store_true", dest="half_sync", default=True,
                          help="synchronisation symbol with two equal halves (even bins only) [default]")
        expert.add_option("", "--full-sync", action="store_false", dest="half_sync", default=True,
                          help="full synchronisation symbol, sent twice")


class ofdm_mapper(object):
    """raw.ofdm_mapper(carrier_map) (rawofdm/src/lib/ofdm/raw_ofdm.i:18-20,
    rawofdm/src/lib/ofdm/raw_ofdm_mapper.cc:22-80): per OFDM symbol the data values go to the
    carriers marked 1 in ascending order, the carriers marked 2 carry +1, -1, +1, ... (the same in
    every symbol), everything else is zero.  In ofdm_mod the block is fused into the modulator
    kernel; on its own, `work` returns its output -- the kernel's `mapped` tap."""

    def __init__(self, carrier_map, ctx=None, device=0):
        self.carrier_map = [int(c) for c in carrier_map]
        if any(c not in (0, 1, 2) for c in self.carrier_map):        # raw_ofdm_mapper.cc:37-38
            raise RuntimeError("raw_ofdm_mapper: carrier_map must include only 0,1,2")
        self.fft_length = len(self.carrier_map)
        self.data_tones = self.carrier_map.count(1)
        self._ctx, self._device = ctx, device

    def work(self, data, enable=None):
        """data: n x data_tones complex64 (tensor or array).  enable (optional, one flag per symbol):
        the block skips symbols whose flag is 0 and leaves their output as it was
        (raw_ofdm_mapper.cc:50-54) -- returned as zeros here.  Returns an n x fft_length tensor."""
        if self._ctx is None:
            self._ctx = TxContext(self._device)
        ctx = self._ctx
        if not torch.is_tensor(data):
            data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.complex64))
        data = data.to(ctx.device).contiguous().reshape(-1)
        n_sym = data.numel() // max(self.data_tones, 1) if self.data_tones else 0
        if self.data_tones and data.numel() != n_sym * self.data_tones:
            raise ValueError("data does not hold whole symbols of %d values" % self.data_tones)
        # no preamble symbols: the tap of a plan whose frames start nowhere but at symbol 0
        flags = np.zeros(max(n_sym, 1), np.uint8)
        flags[0] = 1
        plan = plan_flags(ctx, self.fft_length, 0, self.carrier_map, np.zeros((0, self.fft_length), np.complex64),
                          flags[:n_sym])
        _, mapped = plan.modulate(data, want_mapped=True)
        if enable is not None:
            keep = torch.as_tensor(np.asarray(enable, dtype=np.uint8) != 0, device=ctx.device)
            mapped = mapped * keep[:, None]
        return mapped


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

    def taps(self, mapped, iq):
        """The tap points behind the mapper (rawofdm/src/python/raw_ofdm_mod.py:58-66) of a finished
        modulate(..., want_mapped=True): {"pre": n_rows x N, "ifft": n_rows x N, "cp": n_samples}."""
        dev = self.ctx.device
        pre = torch.zeros((self.n_rows, self.fft_length), dtype=torch.complex64, device=dev)
        ifft = torch.empty((self.n_rows, self.fft_length), dtype=torch.complex64, device=dev)
        cp = torch.empty(self.n_samples, dtype=torch.complex64, device=dev)
        _lib.check(self._lib.ofdmtx_raw_taps(self.ctx._h, self._h, _ptr(mapped), _ptr(iq), _ptr(pre), _ptr(ifft),
                                             _ptr(cp), _stream(dev)))
        return {"pre": pre, "ifft": ifft, "cp": cp}


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

    def _finish(self, plan, data, out=None):
        """out: None, or a function n_samples -> complex64 buffer (or None) for the plan's IQ."""
        if out is not None:
            out = out(plan.n_samples)
        if self._log:                                               # raw_ofdm_mod.py:54-66
            iq, mapped = plan.modulate(data, want_mapped=True, out=out)
            mapped.cpu().numpy().tofile("tx-map.dat")
            for name, t in plan.taps(mapped, iq).items():
                t.cpu().numpy().tofile("tx-%s.dat" % name)
            return iq
        return plan.modulate(data, out=out)

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

    def modulate_frames(self, data, frame_size, out=None):
        """out: see _finish (where the IQ goes; default: a fresh allocation)."""
        p = self.params
        data = self._as_device(data)
        n_sym = data.numel() // p.data_tones
        if n_sym % frame_size:
            raise ValueError("data does not hold whole frames")
        return self._finish(self._plan_frames(n_sym // frame_size, frame_size), data, out)

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
