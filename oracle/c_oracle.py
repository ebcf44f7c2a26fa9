"""ctypes front end of oracle/libftw_oracle.so (CPU ORACLE -- test infrastructure only).

Feeds the C restatement with the reference's literal tables from the committed fixture and
exposes per-packet and batch entry points returning numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import ftw_oracle as _py

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libftw_oracle.so")


class _Tables(C.Structure):
    _fields_ = [("scr", C.c_uint8 * 127), ("pol", C.c_float * 127),
                ("cons", C.c_float * 172), ("preamble", C.c_float * 640)]


def build():
    """Compile the C oracle in-tree (gcc)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libftw_oracle.so"])


_lib = None
_tab = None


def lib():
    global _lib, _tab
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u8p, i32p, i64p, f32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_float), C.POINTER(C.c_double))
    L.orc_packet_sizes.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
    L.orc_packet_sizes.restype = C.c_int
    L.orc_crc32.argtypes = [C.c_char_p, C.c_size_t]
    L.orc_crc32.restype = C.c_uint32
    L.orc_encode_bits.argtypes = [C.POINTER(_Tables), C.c_char_p, C.c_int, C.c_int, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_encode_bits.restype = C.c_int
    L.orc_modulate.argtypes = [C.POINTER(_Tables), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_modulate.restype = C.c_int
    L.orc_tx_batch.argtypes = [C.POINTER(_Tables), C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_float, C.c_int, C.c_double, C.c_int]
    L.orc_tx_batch.restype = C.c_int

    t = _py.tables()
    tab = _Tables()
    tab.scr[:] = [int(v) for v in t.scrambling_seq]
    tab.pol[:] = [float(v) for v in t.pilot_polarity]
    cons = np.concatenate([t.constellation[m] for m in ("bpsk", "qpsk", "qam16", "qam64")])
    ri = np.empty(172, dtype=np.float32)
    ri[0::2], ri[1::2] = cons.real, cons.imag
    tab.cons[:] = ri.tolist()
    ri = np.empty(640, dtype=np.float32)
    ri[0::2], ri[1::2] = t.fft_preamble.real, t.fft_preamble.imag
    tab.preamble[:] = ri.tolist()
    _lib, _tab = L, tab
    return L


def packet_sizes(regime, payload_len):
    L = lib()
    v = [C.c_int() for _ in range(4)]
    rc = L.orc_packet_sizes(int(regime), int(payload_len), *[C.byref(x) for x in v])
    if rc:
        raise ValueError("orc_packet_sizes rc=%d" % rc)
    return {"N_sym": v[0].value, "N_pad": v[1].value, "plcp_len": v[2].value, "msg_len": v[3].value}


def crc32(data):
    return lib().orc_crc32(bytes(data), len(data))


def encode_bits(payload, regime, symboltime=4.0):
    L = lib()
    sz = packet_sizes(regime, len(payload))
    bufs = [np.zeros(sz["plcp_len"], np.uint8), np.zeros(sz["plcp_len"], np.uint8),
            np.zeros(sz["msg_len"], np.uint8), np.zeros(sz["msg_len"], np.uint8)]
    rc = L.orc_encode_bits(C.byref(_tab), bytes(payload), len(payload), int(regime), float(symboltime),
                           *[b.ctypes.data for b in bufs])
    if rc != sz["msg_len"]:
        raise RuntimeError("orc_encode_bits rc=%d" % rc)
    return {"plcp": bufs[0].tobytes(), "scrambled": bufs[1].tobytes(), "coded": bufs[2].tobytes(),
            "interleaved": bufs[3].tobytes(), "N_sym": sz["N_sym"]}


def modulate(msg, regime, pilot_offset=0, amp=0.3, flags=3):
    L = lib()
    regime = int(regime)
    n_cbps = (48, 48, 96, 96, 192, 192, 288, 288)[regime - 1]
    n_sym = (len(msg) - 6) * 8 // n_cbps
    rows = ((6 if flags & 24 else 4) if flags & 1 else 0) + 1 + n_sym + (13 if flags & 2 else 0)
    mapped = np.zeros((1 + n_sym, 48), np.complex64)
    iq = np.zeros(rows * 80, np.complex64)
    iq64 = np.zeros(rows * 80, np.complex128)
    rc = L.orc_modulate(C.byref(_tab), bytes(msg), len(msg), regime, int(pilot_offset), float(amp),
                        int(flags), mapped.ctypes.data, iq.ctypes.data, iq64.ctypes.data)
    if rc != rows * 80:
        raise RuntimeError("orc_modulate rc=%d" % rc)
    return {"mapped": mapped, "iq": iq, "iq64": iq64}


def tx_batch(payloads, pay_off, pay_len, regimes, pilot_off, msg_off, iq_off, msg_total, iq_total,
             amp=0.3, flags=3, symboltime=4.0, threads=1, want_msg=True, want_iq=True):
    """payloads: uint8 array; offsets int64; returns (msg uint8[msg_total] | None, iq complex64[iq_total] | None)."""
    L = lib()
    payloads = np.ascontiguousarray(payloads, np.uint8)
    pay_off = np.ascontiguousarray(pay_off, np.int64)
    pay_len = np.ascontiguousarray(pay_len, np.int32)
    regimes = np.ascontiguousarray(regimes, np.uint8)
    pilot_off = np.ascontiguousarray(pilot_off, np.int32)
    msg_off = np.ascontiguousarray(msg_off, np.int64)
    iq_off = np.ascontiguousarray(iq_off, np.int64)
    msg = np.zeros(msg_total, np.uint8) if want_msg else None
    iq = np.zeros(iq_total, np.complex64) if want_iq else None
    rc = L.orc_tx_batch(C.byref(_tab), len(pay_len), payloads.ctypes.data, pay_off.ctypes.data,
                        pay_len.ctypes.data, regimes.ctypes.data, pilot_off.ctypes.data,
                        msg_off.ctypes.data, iq_off.ctypes.data,
                        msg.ctypes.data if want_msg else None, iq.ctypes.data if want_iq else None,
                        float(amp), int(flags), float(symboltime), int(threads))
    if rc:
        raise RuntimeError("orc_tx_batch rc=%d" % rc)
    return msg, iq
