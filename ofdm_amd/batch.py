"""Batched front end of libofdmtx: thousands to millions of packets per launch.

PyTorch is used only to own device buffers and streams; all arithmetic happens in the HIP
kernels behind the C ABI (ofdm_amd/_lib.py).  Names follow the reference's domain: packets,
PLCP bytes, mapper messages, OFDM symbols, IQ samples.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

SYM_LEN = 80


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class TxContext:
    """One per GPU: owns the device-side tables (scrambler words, constellations, preamble,
    twiddles)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OfdmTxError(_lib.EHIP, "no GPU visible; libofdmtx has no CPU path")
        self.device = torch.device("cuda", int(device))
        h = C.c_void_p()
        _lib.check(self._lib.ofdmtx_create(int(device), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ofdmtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_constellation(self, n_bpsc, points):
        """`constellation` argument of ftw.ofdm_mapper for DATA symbols with n_bpsc bits/carrier."""
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.complex64))
        _lib.check(self._lib.ofdmtx_set_constellation(self._h, int(n_bpsc), pts.ctypes.data, len(pts)))

    def plan_uniform(self, n_pkts, regime, payload_len, payload_stride=0, symbol_time=4.0,
                     pilot_mode=_lib.PILOT_REF_STREAM, pilot_start=0,
                     flags=_lib.F_PREAMBLE | _lib.F_ZEROGAP, amp=0.3):
        h = C.c_void_p()
        _lib.check(self._lib.ofdmtx_plan_uniform(self._h, int(n_pkts), int(regime), int(payload_len),
                                                 int(payload_stride), float(symbol_time), int(pilot_mode),
                                                 int(pilot_start), int(flags), float(amp), C.byref(h)))
        return TxPlan(self, h)

    def plan_ragged(self, regimes, payload_len, payload_off=None, symbol_time=4.0,
                    pilot_mode=_lib.PILOT_REF_STREAM, pilot_start=0,
                    flags=_lib.F_PREAMBLE | _lib.F_ZEROGAP, amp=0.3):
        """Ragged / mixed-rate plan.  numpy arrays (or lists) are walked on the host; torch tensors
        on this context's device (uint8 regimes, int32 lengths, optional int64 offsets) are turned
        into a plan by kernels, without the per-packet host loop."""
        if torch.is_tensor(regimes) and regimes.is_cuda:
            return self._plan_ragged_device(regimes, payload_len, payload_off, symbol_time, pilot_mode,
                                            pilot_start, flags, amp)
        regimes = np.ascontiguousarray(regimes, dtype=np.uint8)
        payload_len = np.ascontiguousarray(payload_len, dtype=np.uint32)
        if len(regimes) != len(payload_len):
            raise ValueError("regimes and payload_len differ in length")
        off = None
        if payload_off is not None:
            off = np.ascontiguousarray(payload_off, dtype=np.uint64)
        h = C.c_void_p()
        _lib.check(self._lib.ofdmtx_plan_ragged(self._h, len(regimes), regimes.ctypes.data,
                                                payload_len.ctypes.data,
                                                off.ctypes.data if off is not None else None,
                                                float(symbol_time), int(pilot_mode), int(pilot_start),
                                                int(flags), float(amp), C.byref(h)))
        return TxPlan(self, h)


    def _plan_ragged_device(self, regimes, payload_len, payload_off, symbol_time, pilot_mode, pilot_start, flags, amp):
        dev = self.device
        if regimes.dtype != torch.uint8 or regimes.device != dev or not regimes.is_contiguous():
            raise ValueError("regimes must be a contiguous uint8 tensor on %s" % dev)
        if (not torch.is_tensor(payload_len) or payload_len.dtype != torch.int32 or payload_len.device != dev or
                not payload_len.is_contiguous() or payload_len.numel() != regimes.numel()):
            raise ValueError("payload_len must be a contiguous int32 tensor on %s, one entry per packet" % dev)
        if payload_off is not None and (not torch.is_tensor(payload_off) or payload_off.dtype != torch.int64 or
                                        payload_off.device != dev or not payload_off.is_contiguous() or
                                        payload_off.numel() != regimes.numel()):
            raise ValueError("payload_off must be a contiguous int64 tensor on %s, one entry per packet" % dev)
        h = C.c_void_p()
        _lib.check(self._lib.ofdmtx_plan_ragged_device(self._h, regimes.numel(), _ptr(regimes), _ptr(payload_len),
                                                       _ptr(payload_off), float(symbol_time), int(pilot_mode),
                                                       int(pilot_start), int(flags), float(amp), _stream(dev),
                                                       C.byref(h)))
        return TxPlan(self, h)


class TxPlan:
    """Geometry of one batch (see include/ofdmtx.h, "plans")."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        s = _lib.PlanSizes()
        _lib.check(self._lib.ofdmtx_plan_get_sizes(self._h, C.byref(s)))
        self.n_pkts = int(s.n_pkts)
        self.payload_bytes = int(s.payload_bytes)
        self.plcp_bytes = int(s.plcp_bytes)
        self.msg_bytes = int(s.msg_bytes)
        self.n_symbols = int(s.n_symbols)
        self.n_rows = int(s.n_rows)
        self.n_samples = int(s.n_samples)
        self._layout = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ofdmtx_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layout(self):
        """Per-packet offsets/lengths as numpy arrays."""
        if self._layout is None:
            n = self.n_pkts
            a = {k: np.zeros(n, np.uint64) for k in ("plcp_off", "msg_off", "sym_off", "sample_off")}
            b = {k: np.zeros(n, np.uint32) for k in ("n_sym", "plcp_len", "msg_len", "pilot_base")}
            _lib.check(self._lib.ofdmtx_plan_get_layout(
                self._h, a["plcp_off"].ctypes.data, a["msg_off"].ctypes.data, a["sym_off"].ctypes.data,
                a["sample_off"].ctypes.data, b["n_sym"].ctypes.data, b["plcp_len"].ctypes.data,
                b["msg_len"].ctypes.data, b["pilot_base"].ctypes.data))
            a.update(b)
            self._layout = a
        return self._layout

    # ---- buffers
    def _u8(self, n):
        return torch.empty(n, dtype=torch.uint8, device=self.ctx.device)

    def alloc_msg(self):
        return self._u8(self.msg_bytes)

    def alloc_iq(self):
        return torch.empty(self.n_samples, dtype=torch.complex64, device=self.ctx.device)

    def _check_payloads(self, payloads):
        if payloads.dtype != torch.uint8 or payloads.device != self.ctx.device or not payloads.is_contiguous():
            raise ValueError("payloads must be a contiguous uint8 tensor on %s" % self.ctx.device)
        if payloads.numel() < self.payload_bytes:
            raise ValueError("payload buffer holds %d bytes, plan reads %d" % (payloads.numel(), self.payload_bytes))

    # ---- kernels
    def encode_bits(self, payloads, taps=False, out=None):
        """Bit chain of send_pkt for the batch.  Returns the mapper-message buffer, or a dict
        with the plcp / scrambled / coded taps as well when taps=True."""
        self._check_payloads(payloads)
        msg = out if out is not None else self.alloc_msg()
        if msg.numel() < self.msg_bytes:
            raise ValueError("message buffer too small")
        plcp = scr = cod = None
        if taps:
            plcp, scr, cod = self._u8(self.plcp_bytes), self._u8(self.plcp_bytes), self._u8(self.msg_bytes)
        _lib.check(self._lib.ofdmtx_encode_bits(self.ctx._h, self._h, _ptr(payloads), _ptr(plcp), _ptr(scr),
                                                _ptr(cod), _ptr(msg), _stream(self.ctx.device)))
        if taps:
            return {"plcp": plcp, "scrambled": scr, "coded": cod, "interleaved": msg}
        return msg

    def modulate(self, msg, want_mapped=False, out=None):
        """Mapper messages -> IQ stream (and the mapper block's symbols when want_mapped)."""
        if msg.dtype != torch.uint8 or msg.device != self.ctx.device or msg.numel() < self.msg_bytes:
            raise ValueError("message buffer must be uint8 on %s with >= %d bytes" % (self.ctx.device, self.msg_bytes))
        iq = out if out is not None else self.alloc_iq()
        if iq.numel() < self.n_samples or iq.dtype != torch.complex64:
            raise ValueError("iq buffer must be complex64 with >= %d samples" % self.n_samples)
        mapped = None
        if want_mapped:
            mapped = torch.empty((self.n_symbols, 48), dtype=torch.complex64, device=self.ctx.device)
        _lib.check(self._lib.ofdmtx_modulate(self.ctx._h, self._h, _ptr(msg), _ptr(mapped), _ptr(iq),
                                             _stream(self.ctx.device)))
        return (iq, mapped) if want_mapped else iq

    def tx(self, payloads, out=None, work=None):
        """payload -> IQ for the whole batch."""
        self._check_payloads(payloads)
        iq = out if out is not None else self.alloc_iq()
        # (the fused kernel keeps the mapper messages in LDS: no work buffer is needed)
        _lib.check(self._lib.ofdmtx_tx(self.ctx._h, self._h, _ptr(payloads), _ptr(work), _ptr(iq),
                                       _stream(self.ctx.device)))
        return iq

    # ---- loop-back verifier
    def rx_demod(self, iq, out=None):
        """IQ stream of this plan's geometry -> (hard coded bits in the layout of the `coded` tap,
        per-packet status words)."""
        if iq.dtype != torch.complex64 or iq.device != self.ctx.device or iq.numel() < self.n_samples:
            raise ValueError("iq must be complex64 on %s with >= %d samples" % (self.ctx.device, self.n_samples))
        coded = out if out is not None else self.alloc_msg()
        status = torch.zeros(max(self.n_pkts, 1), dtype=torch.int32, device=self.ctx.device)
        _lib.check(self._lib.ofdmtx_rx_demod(self.ctx._h, self._h, _ptr(iq), _ptr(coded), _ptr(status),
                                             _stream(self.ctx.device)))
        return coded, status[:self.n_pkts]

    def rx_decode(self, coded, status=None):
        """Coded bits -> (payload bytes at the plan's payload offsets, status words; bits RX_*)."""
        if coded.dtype != torch.uint8 or coded.device != self.ctx.device or coded.numel() < self.msg_bytes:
            raise ValueError("coded buffer must be uint8 on %s with >= %d bytes" % (self.ctx.device, self.msg_bytes))
        if status is None:
            status = torch.zeros(max(self.n_pkts, 1), dtype=torch.int32, device=self.ctx.device)[:self.n_pkts]
        payloads = torch.zeros(max(self.payload_bytes, 1), dtype=torch.uint8, device=self.ctx.device)
        _lib.check(self._lib.ofdmtx_rx_decode(self.ctx._h, self._h, _ptr(coded), _ptr(payloads), _ptr(status),
                                              _stream(self.ctx.device)))
        return payloads[:self.payload_bytes], status

    def loopback(self, iq):
        """Receive what tx() / modulate() produced: (payload bytes, status words -- all zero and
        equal to the transmitted payloads for a clean loop-back)."""
        coded, status = self.rx_demod(iq)
        return self.rx_decode(coded, status)
