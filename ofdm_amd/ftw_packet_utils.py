"""Drop-in for the reference's ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py (cited as pu).

Same function names, arguments and return values; byte strings are `bytes` (the reference is
Python 2, where str is bytes).  The bit-level work of every function runs on the GPU through
libofdmtx: ftw_make through the batched kernel's PLCP tap (a batch of one), scrambler /
conv_encoder / interleaver through the single-message stage operators.  There is no CPU path:
without the library or a GPU these functions raise.

The per-packet functions exist for API compatibility; throughput comes from batches
(ofdm_amd.batch, ofdm_amd.ftw_ofdm.ofdm_mod).
"""
import ctypes as C
import struct

import numpy as np
import torch

from . import _lib
from . import ftw

MAXLEN = 3168                                   # pu:119
VERBOSE = False                                 # the reference prints four lines per ftw_make (pu:123-126)

_MODULATION = {"1": "bpsk", "2": "bpsk", "3": "qpsk", "4": "qpsk", "5": "qam16", "6": "qam16",
               "7": "qam64", "8": "qam64"}
_RATE = {"1": 1 / float(2), "2": 3 / float(4), "3": 1 / float(2), "4": 3 / float(4), "5": 1 / float(2),
         "6": 3 / float(4), "7": 2 / float(3), "8": 3 / float(4)}

_ctx = {}


def _context(device=None):
    """One library context per device, created on first use."""
    from .batch import TxContext
    device = torch.cuda.current_device() if device is None and torch.cuda.is_available() else (device or 0)
    if device not in _ctx:
        _ctx[device] = TxContext(device)
    return _ctx[device]


def _as_bytes(s):
    if isinstance(s, str):
        return s.encode("latin-1")
    return bytes(s)


def _regime(regime):
    r = str(regime)
    if r not in _MODULATION:
        # pu:39-78 has no else branch: an unknown regime leaves `modulation` unbound
        raise UnboundLocalError("local variable 'modulation' referenced before assignment")
    return r


def get_info(payload, regime, symboltime):
    """pu:25-105."""
    payload = _as_bytes(payload)
    r = _regime(regime)
    mac_framectrl = b"\x08\x00"
    mac_address1 = b"\x00\x60\x08\xcd\x37\xa6"
    mac_address2 = b"\x00\x20\xd6\x01\x3c\xf1"
    mac_address3 = b"\x00\x60\x08\xad\x3b\xaf"
    mac_seqctrl = b"\x00\x00"
    info = _lib.PktInfo()
    # the length check belongs to ftw_make (pu:119-121); get_info itself accepts any length
    rc = _lib.load().ofdmtx_packet_info(int(r), len(payload), float(symboltime), 0, C.byref(info))
    if rc != _lib.ELENGTH:
        _lib.check(rc)
    vals = {k: getattr(info, k) for k in ("n_cbps", "n_bpsc", "n_dbps", "n_sym", "n_data", "n_pad", "txtime")}
    txtime = vals["txtime"]
    mac_duration = bytes([(txtime >> 8) & 0xff, txtime & 0xff])
    packet = mac_framectrl + mac_duration + mac_address1 + mac_address2 + mac_address3 + mac_seqctrl + payload
    return {"packet": packet, "packet_len": len(packet) + 4, "txtime": txtime, "mac_duration": mac_duration,
            "modulation": _MODULATION[r], "rate": _RATE[r], "N_sym": vals["n_sym"], "N_data": vals["n_data"],
            "N_pad": vals["n_pad"], "N_cbps": vals["n_cbps"], "N_bpsc": vals["n_bpsc"], "N_dbps": vals["n_dbps"]}


def ftw_make(payload, regime, symboltime):
    """pu:108-205: (PLCP_HEADER + MPDU_with_crc32 + TAIL_and_PAD, Length)."""
    payload = _as_bytes(payload)
    r = _regime(regime)
    info = get_info(payload, r, symboltime)
    if info["packet_len"] > MAXLEN:
        raise ValueError("MPDU-length must be in [0, %d]" % (MAXLEN,))
    if VERBOSE:
        print("Txtime in microseconds:", info["txtime"])
        print("Length of MPDU in bytes = ", info["packet_len"])
        print("Number of DATA OFDM symbols = ", info["N_sym"])
        print("Number of padding bits = ", info["N_pad"])
    ctx = _context()
    plan = ctx.plan_uniform(1, int(r), len(payload), symbol_time=float(symboltime))
    dev_payload = torch.frombuffer(bytearray(payload) or bytearray(1), dtype=torch.uint8).to(ctx.device)
    taps = plan.encode_bits(dev_payload, taps=True)
    n = int(plan.layout()["plcp_len"][0])
    return taps["plcp"][:n].cpu().numpy().tobytes(), info["packet_len"]


def make_MPDU(payload):
    """pu:208-218: bit order reversed inside every byte (host helper, a 256-entry table)."""
    return _as_bytes(payload).translate(_BITREV8)


_BITREV8 = bytes(int("{:08b}".format(i)[::-1], 2) for i in range(256))


def _stage(fn_name, data, out_len, *args):
    ctx = _context()
    data = _as_bytes(data)
    src = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).to(ctx.device)
    dst = torch.empty(max(out_len, 1), dtype=torch.uint8, device=ctx.device)
    fn = getattr(ctx._lib, fn_name)
    stream = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
    rc = fn(ctx._h, C.c_void_p(src.data_ptr()), len(data), *args, C.c_void_p(dst.data_ptr()), stream)
    return rc, dst, out_len


def scrambler(pkt, Length_data):
    """pu:344-367: (scrambled bytes, Length_data)."""
    rc, dst, n = _stage("ofdmtx_scramble", pkt, len(_as_bytes(pkt)), int(Length_data))
    if rc == _lib.EINVAL:
        raise IndexError("list assignment index out of range")      # pu:361-362 past the end of the string
    _lib.check(rc)
    return dst[:n].cpu().numpy().tobytes(), Length_data


def conv_encoder(pkt, Length, regime, N_cbps, N_bpsc, N_sym, N_rate):
    """pu:280-341: coded and punctured bytes (48 SIGNAL bits + N_sym * N_cbps bits)."""
    r = _regime(regime)
    out_len = 6 + int(N_sym) * int(N_cbps) // 8
    rc, dst, n = _stage("ofdmtx_conv_encode", pkt, out_len, int(r), int(N_sym))
    if rc == _lib.EINVAL:
        raise ValueError("conv_encoder: input too short for N_sym = %d symbols" % N_sym)
    _lib.check(rc)
    return dst[:n].cpu().numpy().tobytes()


def interleaver(payload, regime, N_cbps, N_bpsc):
    """pu:222-277."""
    r = _regime(regime)
    rc, dst, n = _stage("ofdmtx_interleave", payload, len(_as_bytes(payload)), int(r))
    if rc == _lib.EINVAL:
        raise ValueError("interleaver: input must hold the SIGNAL symbol and whole DATA symbols")
    _lib.check(rc)
    return dst[:n].cpu().numpy().tobytes()


def gen_and_append_crc32(MPDU, packet_for_crc):
    """pu:370-372."""
    crc = ftw.ftw_crc32(packet_for_crc)
    return _as_bytes(MPDU) + struct.pack(">I", crc & 0xFFFFFFFF)


def insert_preamble(length, N_sym, role=None):
    """pu:374-383."""
    ftw_preamble = [list(fft_preamble)]
    if role == 'A':
        return ftw.pnc_ofdm_preamble(length, N_sym, ftw_preamble, 1)
    if role == 'B':
        return ftw.pnc_ofdm_preamble(length, N_sym, ftw_preamble, 2)
    return ftw.ofdm_preamble(length, N_sym, ftw_preamble)


def insert_zerogap(length, N_sym, state='FTW'):
    """pu:385-392."""
    gap = [list(gap_sample)]
    if state == 'FPNC':
        return ftw.pnc_zerogap(length, N_sym, gap)
    return ftw.zerogap(length, N_sym, gap)


def qam16(self):
    return gray_const_qam16


def qam64(self):
    return gray_const_qam64


def qpsk(self):
    return gray_const_qpsk


def bpsk(self):
    return gray_const_bpsk


def conv_packed_binary_string_to_1_0_string(s):
    """pu:406-417: b'\\xAF' --> '10101111'."""
    return "".join("{:08b}".format(b) for b in _as_bytes(s))


def conv_1_0_string_to_packed_binary_string(s):
    """pu:419-448: '10101111' -> b'\\xAF' (left-padded with zeros to a multiple of 8)."""
    if not is_1_0_string(s):
        raise ValueError("Input must be a string containing only 0's and 1's")
    rem = len(s) % 8
    if rem != 0:
        s = '0' * (8 - rem) + s
    return bytes(int(s[i:i + 8], 2) for i in range(0, len(s), 8))


def is_1_0_string(s):
    """pu:451-457."""
    if not isinstance(s, str):
        return False
    return all(ch in ('0', '1') for ch in s)


def string_to_hex_list(s):
    """pu:459-460."""
    return [hex(b) for b in _as_bytes(s)]


def hexint(mask):
    """pu:463-473: unsigned mask -> signed int."""
    if mask >= 2 ** 31:
        return int(mask - 2 ** 32)
    return mask


def ascii_to_bin(char):
    """pu:476-490."""
    return "{:08b}".format(ord(char) if isinstance(char, str) else int(char[0]))


def _table(which):
    return tuple(complex(v) for v in _lib.get_table(which))


# Static tables (pu:497-510), served by the library so that host and device hold the same float32 values.
try:
    gray_const_bpsk = _table(_lib.TABLE_BPSK)
    gray_const_qpsk = _table(_lib.TABLE_QPSK)
    gray_const_qam16 = _table(_lib.TABLE_QAM16)
    gray_const_qam64 = _table(_lib.TABLE_QAM64)
    fft_preamble = _table(_lib.TABLE_PREAMBLE)
except _lib.OfdmTxError:                        # library not built: importing stays possible, calls raise
    gray_const_bpsk = gray_const_qpsk = gray_const_qam16 = gray_const_qam64 = fft_preamble = ()
gap_sample = (0j,) * 80                         # pu:510
