"""CPU ORACLE (test infrastructure, not product code) -- numpy restatement of the FTW 802.11a TX chain.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It restates, stage by stage, what the reference computes for one packet; every
function cites the reference lines it follows (paths relative to the reference root).
The reference itself is never imported or executed: it is Python-2/GNU-Radio code and
running it in any form was denied in this environment, so this restatement is pinned by

  * the literal tables of the reference (tests/golden/ftw_tables.json, extracted as text),
  * the 802.11-2007 Annex G.8 time-domain frame the reference ships
    (ftw_80211_ofdm_tx/src/matlab/802.11-2007-Annex-G.8.txt) -- end to end, ~1e-3 abs,
  * structural identities (CRC = bit-reversed zlib CRC, LFSR sequences, permutation checks).

Bit-level results beyond what Annex G.8 exercises (regime 6) are "parity unpinned" by any
reference-held vector; IQ at 1e-5 is pinned against the mathematically defined transform
(float64 inverse DFT), not against FFTW3f output, which the reference does not ship.

Conventions: packed bytes are MSB-first (ftw_packet_utils.py:406-417); "bits" arrays are
uint8 arrays of 0/1 in transmit order.
"""
import json
import math
import os
import zlib

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES_JSON = os.path.join(_HERE, "..", "tests", "golden", "ftw_tables.json")

MAXLEN = 3168          # ftw_packet_utils.py:119
FFT_LEN = 64           # ftw_ofdm.py:70
CP_LEN = 16            # ftw_ofdm.py:73
SYM_LEN = 80           # ftw_ofdm.py:75
N_DATA_CARRIERS = 48   # ftw_ofdm.py:72
N_PREAMBLE_ROWS = 4    # ftw_ofdm_preamble.cc:79
N_GAP_ROWS = 13        # ftw_zerogap.cc:88

# regime -> (modulation, (rate num, rate den), N_cbps, N_bpsc, RATE nibble)
# ftw_packet_utils.py:39-78 and :129-144
REGIMES = {
    "1": ("bpsk", (1, 2), 48, 1, 0x0D),
    "2": ("bpsk", (3, 4), 48, 1, 0x0F),
    "3": ("qpsk", (1, 2), 96, 2, 0x05),
    "4": ("qpsk", (3, 4), 96, 2, 0x07),
    "5": ("qam16", (1, 2), 192, 4, 0x09),
    "6": ("qam16", (3, 4), 192, 4, 0x0B),
    "7": ("qam64", (2, 3), 288, 6, 0x01),
    "8": ("qam64", (3, 4), 288, 6, 0x03),
}

# MAC header of get_info (ftw_packet_utils.py:28-36), duration filled in later
_MAC_FC = bytes([0x08, 0x00])
_MAC_A1 = bytes([0x00, 0x60, 0x08, 0xCD, 0x37, 0xA6])
_MAC_A2 = bytes([0x00, 0x20, 0xD6, 0x01, 0x3C, 0xF1])
_MAC_A3 = bytes([0x00, 0x60, 0x08, 0xAD, 0x3B, 0xAF])
_MAC_SEQ = bytes([0x00, 0x00])


class Tables:
    """Literal tables of the reference, loaded from the committed fixture."""

    def __init__(self, path=_TABLES_JSON):
        with open(path) as f:
            t = json.load(f)
        c = lambda rows: np.array([complex(r, i) for r, i in rows], dtype=np.complex128)
        self.scrambling_seq = np.array(t["scrambling_seq"], dtype=np.uint8)      # :346-349
        self.pilot_polarity = np.array(t["pilot_polarity"], dtype=np.float64)    # pilot_cc.cc:96-104
        self.constellation = {                                                   # :497-506
            "bpsk": c(t["gray_const_bpsk"]), "qpsk": c(t["gray_const_qpsk"]),
            "qam16": c(t["gray_const_qam16"]), "qam64": c(t["gray_const_qam64"]),
        }
        # the preamble block stores its table as complex64 (std::vector<gr_complex>), so the
        # values that reach the output are the float32 roundings of the literals
        self.fft_preamble = c(t["fft_preamble"]).astype(np.complex64).astype(np.complex128)   # :508
        self.gap_sample = c(t["gap_sample"])                                     # :510
        self.annex_g8_time = c(t["annex_g8_time"])


_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = Tables()
    return _tables


# --------------------------------------------------------------------------- helpers

def bytes_to_bits(b):
    """ftw_packet_utils.py:406-417 -- MSB-first expansion."""
    return np.unpackbits(np.frombuffer(bytes(b), dtype=np.uint8))


def bits_to_bytes(bits):
    """ftw_packet_utils.py:419-448 -- MSB-first packing; left-pads with zeros to a byte multiple."""
    bits = np.asarray(bits, dtype=np.uint8)
    rem = len(bits) % 8
    if rem:
        bits = np.concatenate([np.zeros(8 - rem, dtype=np.uint8), bits])
    return np.packbits(bits).tobytes()


def _bitrev8(b):
    return np.packbits(np.unpackbits(np.frombuffer(bytes(b), dtype=np.uint8)).reshape(-1, 8)[:, ::-1]).tobytes()


# --------------------------------------------------------------------------- a1: get_info

def get_info(payload, regime, symboltime):
    """ftw_packet_utils.py:25-105."""
    payload = bytes(payload)
    modulation, (rn, rd), n_cbps, n_bpsc, _ = REGIMES[regime]      # KeyError for unknown regimes
    rate = rn / float(rd)
    packet_wo_dur_len = 24 + len(payload)
    n_dbps = int(n_cbps * rate)                                    # :81
    nbits = 16 + 32 + 6 + 8 * packet_wo_dur_len
    n_sym = int(math.ceil(nbits / float(n_dbps)))                  # :85
    txtime = int(5 * symboltime + symboltime * n_sym) + 2          # :90
    mac_duration = bytes([(txtime >> 8) & 0xFF, txtime & 0xFF])    # :91
    n_data = int(n_sym * n_dbps)                                   # :94
    n_pad = n_data - nbits                                         # :97
    packet = _MAC_FC + mac_duration + _MAC_A1 + _MAC_A2 + _MAC_A3 + _MAC_SEQ + payload   # :100
    return {"packet": packet, "packet_len": len(packet) + 4, "txtime": txtime,
            "mac_duration": mac_duration, "modulation": modulation, "rate": rate,
            "N_sym": n_sym, "N_data": n_data, "N_pad": n_pad, "N_cbps": n_cbps,
            "N_bpsc": n_bpsc, "N_dbps": n_dbps}


# --------------------------------------------------------------------------- a4: CRC

def ftw_crc32(data):
    """ftw_crc32.cc:33-82: reflected CRC-32, state bit-reversed over 32 bits, then ^0xFFFFFFFF.
    Equals bitreverse32(zlib.crc32(data))."""
    c = zlib.crc32(bytes(data)) & 0xFFFFFFFF
    return int("{:032b}".format(c)[::-1], 2)


def ftw_crc32_bitwise(data):
    """Same value, derived without zlib (used to cross-check the identity above)."""
    crc = 0xFFFFFFFF
    for byte in bytes(data):
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0xEDB88320 if crc & 1 else 0)
    rev = 0
    for j in range(32):
        rev |= ((crc >> j) & 1) << (31 - j)
    return rev ^ 0xFFFFFFFF


# --------------------------------------------------------------------------- a2/a3: PLCP build

def signal_field(regime, length):
    """24 SIGNAL bits as 3 bytes -- ftw_packet_utils.py:128-180."""
    rate_bits = REGIMES[regime][4]
    rev = 0
    for i in range(12):                                            # :147-150 LSB first
        rev |= ((length >> i) & 1) << (11 - i)
    parity = (bin(rev & 0xFFF).count("1") + bin(rate_bits).count("1")) % 2   # :153-157
    field = (rate_bits << 20) | (rev << 7) | (parity << 6)         # :176
    return bytes([(field >> 16) & 0xFF, (field >> 8) & 0xFF, field & 0xFF])


def tail_and_pad_bytes(n_pad):
    """ftw_packet_utils.py:160-171: ceil((N_pad+6)/8) zero bytes."""
    return (n_pad + 6 + 7) // 8


def plcp_from_mpdu(mpdu, regime, fcs=None):
    """PLCP byte string for an arbitrary MPDU (header+body, natural byte order).

    Same construction as ftw_make (ftw_packet_utils.py:128-205) but with the MPDU supplied
    by the caller; `fcs` (4 bytes, as appended after the bit-reversed MPDU) overrides the
    computed one -- needed for the Annex G frame, whose FCS is deliberately the one printed
    in the standard (ftw_80211_ofdm_tx/src/matlab/encode.m:221-223).
    Returns (plcp_bytes, Length, N_sym, N_pad)."""
    mpdu = bytes(mpdu)
    _, (rn, rd), n_cbps, _, _ = REGIMES[regime]
    n_dbps = int(n_cbps * (rn / float(rd)))
    length = len(mpdu) + 4
    if length > MAXLEN:
        raise ValueError("MPDU-length must be in [0, %d]" % (MAXLEN,))       # :119-121
    nbits = 16 + 32 + 6 + 8 * len(mpdu)
    n_sym = int(math.ceil(nbits / float(n_dbps)))
    n_pad = n_sym * n_dbps - nbits
    if fcs is None:
        fcs = ftw_crc32(mpdu).to_bytes(4, "big")                   # :370-372, struct.pack(">I")
    out = (signal_field(regime, length) + b"\x00\x00"              # SIGNAL + SERVICE :183-185
           + _bitrev8(mpdu)                                        # make_MPDU :208-218
           + bytes(fcs)
           + b"\x00" * tail_and_pad_bytes(n_pad))
    return out, length, n_sym, n_pad


def ftw_make(payload, regime, symboltime):
    """ftw_packet_utils.py:108-205 (without the four prints)."""
    info = get_info(payload, regime, symboltime)
    pkt, length, _, _ = plcp_from_mpdu(info["packet"], regime)
    return pkt, length


# --------------------------------------------------------------------------- a5: scrambler

def scrambler(pkt, length_data):
    """ftw_packet_utils.py:344-367."""
    bits = bytes_to_bits(pkt).copy()
    seq = tables().scrambling_seq
    n = len(bits) - 24
    k = np.arange(n)
    bits[24:] ^= seq[k % 127]                                      # :357-358
    z = 8 * length_data
    bits[40 + z:46 + z] = 0                                        # :361-362 (indices 23+z+17 .. 23+z+22)
    return bits_to_bytes(bits), length_data


# --------------------------------------------------------------------------- a6: encoder + puncture

def conv_encoder(pkt, length, regime, n_cbps, n_bpsc, n_sym, n_rate):
    """ftw_packet_utils.py:280-341.  One continuous K=7 pass (g0=133o -> A, g1=171o -> B)
    over SIGNAL+DATA, then puncturing of the DATA part only."""
    u = bytes_to_bits(pkt).astype(np.uint8)
    up = np.concatenate([np.zeros(6, dtype=np.uint8), u])
    d = lambda k: up[6 - k:6 - k + len(u)]                         # u[n-k]
    # register = (reg >> 1) | (b << 6): bit6 = u[n], bit0 = u[n-6]  (:292-297)
    # g0 = 0x5B = bits {6,4,3,1,0} -> delays {0,2,3,5,6}; g1 = 0x79 = bits {6,5,4,3,0} -> {0,1,2,3,6}
    a = d(0) ^ d(2) ^ d(3) ^ d(5) ^ d(6)
    b = d(0) ^ d(1) ^ d(2) ^ d(3) ^ d(6)
    enc = np.empty(2 * len(u), dtype=np.uint8)
    enc[0::2] = a
    enc[1::2] = b
    real = n_sym * n_cbps                                          # :282
    sig, data = enc[:48], enc[48:]
    if regime in ("1", "3", "5"):                                  # :310-314
        out = enc[:48 + real]
    elif regime in ("2", "4", "6", "8"):                           # :316-328: keep p%6 in {0,1,2,5}
        p = np.arange(len(data))
        keep = data[np.isin(p % 6, (0, 1, 2, 5))]
        out = np.concatenate([sig, keep[:real]])
    elif regime == "7":                                            # :330-341: keep p%4 in {0,1,2}
        p = np.arange(len(data))
        keep = data[p % 4 != 3]
        out = np.concatenate([sig, keep[:real]])
    else:
        raise KeyError(regime)
    assert len(out) == 48 + real
    return bits_to_bytes(out)


# --------------------------------------------------------------------------- a7: interleaver

def interleave_perm(n_cbps, n_bpsc, signal=False):
    """dest[k] for source position k within one symbol.
    first_permutation ftw_packet_utils.py:231-258, second_permutation :261-277."""
    k = np.arange(n_cbps)
    i = (n_cbps // 16) * (k % 16) + k // 16
    if signal:
        return i
    s = max(n_bpsc // 2, 1)
    j = s * (i // s) + (i + n_cbps - (16 * i) // n_cbps) % s
    return j


def interleaver(pkt, regime, n_cbps, n_bpsc):
    """ftw_packet_utils.py:222-277: SIGNAL symbol (48 bits) gets the first permutation only,
    DATA symbols both."""
    bits = bytes_to_bits(pkt)
    out = np.empty_like(bits)
    out[interleave_perm(48, 1, signal=True)] = bits[:48]
    n_sym = (len(bits) - 48) // n_cbps
    body = bits[48:48 + n_sym * n_cbps].reshape(n_sym, n_cbps)
    dst = np.empty_like(body)
    dst[:, interleave_perm(n_cbps, n_bpsc)] = body
    out[48:48 + n_sym * n_cbps] = dst.reshape(-1)
    return bits_to_bytes(out)


# --------------------------------------------------------------------------- a10: send_pkt bit chain

def encode_bits(payload, regime, symboltime=4.0):
    """ofdm_mod.send_pkt up to the message handed to the mapper -- ftw_ofdm.py:201-217.
    Returns dict with every intermediate byte string."""
    info = get_info(payload, regime, symboltime)
    pkt, length = ftw_make(payload, regime, symboltime)
    scr, _ = scrambler(pkt, length)
    cod = conv_encoder(scr, length, regime, info["N_cbps"], info["N_bpsc"], info["N_sym"], info["rate"])
    itl = interleaver(cod, regime, info["N_cbps"], info["N_bpsc"])
    return {"info": info, "plcp": pkt, "Length": length, "scrambled": scr, "coded": cod, "interleaved": itl}


def encode_bits_from_mpdu(mpdu, regime, fcs=None):
    """Same chain for a caller-supplied MPDU / FCS (Annex G known-answer entry)."""
    _, _, n_cbps, n_bpsc, _ = REGIMES[regime]
    pkt, length, n_sym, _ = plcp_from_mpdu(mpdu, regime, fcs)
    scr, _ = scrambler(pkt, length)
    cod = conv_encoder(scr, length, regime, n_cbps, n_bpsc, n_sym, None)
    itl = interleaver(cod, regime, n_cbps, n_bpsc)
    return {"plcp": pkt, "Length": length, "N_sym": n_sym, "scrambled": scr, "coded": cod, "interleaved": itl}


def encode_bits_from_psdu(psdu, regime):
    """The same chain for a whole PSDU (MPDU + FCS bytes in natural order, each sent least significant
    bit first): what a plan with OFDMTX_F_RAW_PSDU transmits.  LENGTH = len(psdu); the PLCP string is
    built as in ftw_make (ftw_packet_utils.py:128-205) with the bit-reversed PSDU in the place of
    MPDU + FCS.  For four bytes or more this is encode_bits_from_mpdu(psdu[:-4], regime,
    bit-reversed psdu[-4:])."""
    psdu = bytes(psdu)
    _, (rn, rd), n_cbps, n_bpsc, _ = REGIMES[regime]
    n_dbps = int(n_cbps * (rn / float(rd)))
    length = len(psdu)
    if length > MAXLEN:
        raise ValueError("MPDU-length must be in [0, %d]" % (MAXLEN,))
    nbits = 16 + 8 * length + 6
    n_sym = int(math.ceil(nbits / float(n_dbps)))
    n_pad = n_sym * n_dbps - nbits
    pkt = signal_field(regime, length) + b"\x00\x00" + _bitrev8(psdu) + b"\x00" * tail_and_pad_bytes(n_pad)
    scr, _ = scrambler(pkt, length)
    cod = conv_encoder(scr, length, regime, n_cbps, n_bpsc, n_sym, None)
    itl = interleaver(cod, regime, n_cbps, n_bpsc)
    return {"plcp": pkt, "Length": length, "N_sym": n_sym, "scrambled": scr, "coded": cod, "interleaved": itl}


# --------------------------------------------------------------------------- a11: mapper

def ofdm_mapper(msg, regime):
    """ftw_ofdm_mapper.cc:137-297: message bytes -> (1+N_sym) x 48 complex.
    Bytes 0..5 are the SIGNAL symbol, always 1 bit per carrier on the local BPSK table
    {-1,+1} (:179,184-191,238-240); the rest takes N_bpsc bits per carrier MSB first."""
    modulation, _, n_cbps, n_bpsc, _ = REGIMES[regime]
    bits = bytes_to_bits(msg)
    n_sym = (len(bits) - 48) // n_cbps
    assert 48 + n_sym * n_cbps == len(bits), "message is not a whole number of symbols"
    out = np.empty((1 + n_sym, 48), dtype=np.complex64)
    out[0] = np.array([-1.0, 1.0], dtype=np.float32)[bits[:48]]
    body = bits[48:].reshape(n_sym, 48, n_bpsc).astype(np.int64)
    w = 1 << np.arange(n_bpsc - 1, -1, -1)
    idx = (body * w).sum(axis=2)
    out[1:] = tables().constellation[modulation].astype(np.complex64)[idx]
    flags = np.zeros(1 + n_sym, dtype=np.uint8)
    flags[0] = 1                                                   # :291-294 start-of-packet flag
    return out, flags


# --------------------------------------------------------------------------- a12/a13: pilots, carrier map

def ofdm_pilot(sym48, pilot_offset=0, user=None):
    """ftw_ofdm_pilot_cc.cc:86-166: 48 -> 53 carriers; symbol c uses polarity[(c+offset)%127].
    `pilot_offset` models the block's process-global running counter (:82,:159).
    user 1 / 2 / 0: ftw_pnc_ofdm_pilot_cc.cc:125-131,146-178 instead -- the same carrier layout
    with constant pilots (1,0,1,0) / (0,1,0,1) / (1,1,1,1) at -21, -7, +7, +21, no polarity
    sequence and no sign change on the fourth."""
    sym48 = np.asarray(sym48)
    n = sym48.shape[0]
    out = np.zeros((n, 53), dtype=np.complex64)
    if user is None:
        p = tables().pilot_polarity[(np.arange(n) + pilot_offset) % 127].astype(np.float32)
        pv = [p, p, p, -p]
    else:
        pv = {1: [1, 0, 1, 0], 0: [1, 1, 1, 1]}.get(user, [0, 1, 0, 1])      # :127 "if (d_user != 1)"
    out[:, 0:5] = sym48[:, 0:5]
    out[:, 5] = pv[0]
    out[:, 6:19] = sym48[:, 5:18]
    out[:, 19] = pv[1]
    out[:, 20:26] = sym48[:, 18:24]
    out[:, 26] = 0
    out[:, 27:33] = sym48[:, 24:30]
    out[:, 33] = pv[2]
    out[:, 34:47] = sym48[:, 30:43]
    out[:, 47] = pv[3]
    out[:, 48:53] = sym48[:, 43:48]
    return out


def pnc_preamble(user):
    """ftw_pnc_ofdm_preamble.cc:80-103: six rows -- the two short-training rows of the table,
    then its two long-training rows followed (user 1) or preceded (user 2) by two zero rows."""
    rows = tables().fft_preamble.reshape(4, SYM_LEN)
    z = np.zeros((2, SYM_LEN), dtype=rows.dtype)
    if user == 1:
        return np.concatenate([rows[0:2], rows[2:4], z]).reshape(-1)
    if user == 2:
        return np.concatenate([rows[0:2], z, rows[2:4]]).reshape(-1)
    raise ValueError("user must be 1 or 2")                        # :105 prints an error and emits garbage


def ofdm_cmap(sym53):
    """ftw_ofdm_cmap_cc.cc:101-109: out[0..26]=in[26..52], out[27..37]=0, out[38..63]=in[0..25]."""
    sym53 = np.asarray(sym53)
    out = np.zeros((sym53.shape[0], 64), dtype=np.complex64)
    out[:, 0:27] = sym53[:, 26:53]
    out[:, 38:64] = sym53[:, 0:26]
    return out


# --------------------------------------------------------------------------- a14-a16: IFFT, CP, scale

def ifft_unnormalised(bins):
    """gr.fft_vcc(N, False, [], False) -- ftw_ofdm.py:118: x[n] = sum_k X[k] e^{+j2pi kn/N}.
    Evaluated in float64 (the 1e-5 parity bar is against this exact transform)."""
    bins = np.asarray(bins, dtype=np.complex128)
    return np.fft.ifft(bins, axis=-1) * bins.shape[-1]


def cyclic_prefix(x, cp):
    """digital.ofdm_cyclic_prefixer(N, N+cp) -- ftw_ofdm.py:121-123: [x[N-cp..N-1], x]."""
    return np.concatenate([x[..., x.shape[-1] - cp:], x], axis=-1)


# --------------------------------------------------------------------------- a17-a19: framing

def assemble_ppdu(sym80, amp=0.3, swap_iq=False):
    """ftw_ofdm_preamble.cc:79-89 (4x80 table samples in front), ftw_zerogap.cc:81-92
    (13x80 zeros behind), ftw_ofdm_tx_uhd.py:82-86 (x tx_amplitude), ftw_ofdm.py:155-172 (IQ swap)."""
    t = tables()
    s = np.concatenate([t.fft_preamble, np.asarray(sym80).reshape(-1), np.zeros(N_GAP_ROWS * SYM_LEN)])
    s = s * float(np.float32(amp))
    if swap_iq:
        s = s.imag + 1j * s.real
    return s


def modulate(msg, regime, pilot_offset=0, amp=0.3, swap_iq=False, preamble=True, zerogap=True, role=None,
             preamble_table=None, gap_row=None):
    """Mapper message -> per-stage arrays, following the flowgraph of ftw_ofdm.py:175-179.
    role 'A' / 'B': the PNC blocks of ftw_ofdm.py:101-106,133-138,147-149 (pilot and preamble
    blocks with user 1 / 2; ftw_pnc_zerogap.cc:81-92 appends the same 13 zero rows).
    preamble_table (>= 320 samples) / gap_row (80 samples): the block arguments of
    ftw_ofdm_preamble.cc:79-83 (first four rows of row 0 of the table are emitted) and
    ftw_zerogap.cc:88-92 (row 0 is emitted 13 times) when they are not the literals of
    ftw_packet_utils.py:508,510; both are stored as complex64 by the blocks."""
    user = {None: None, "A": 1, "B": 2}[role]
    mapped, flags = ofdm_mapper(msg, regime)
    pil = ofdm_pilot(mapped, pilot_offset, user)
    bins = ofdm_cmap(pil)
    x = ifft_unnormalised(bins)
    cp = cyclic_prefix(x, CP_LEN)
    scaled = cp * (1.0 / math.sqrt(FFT_LEN))                       # ftw_ofdm.py:126
    t = tables()
    parts = []
    if preamble:
        if preamble_table is not None:
            assert user is None
            tab = np.asarray(preamble_table).astype(np.complex64).astype(np.complex128)
            parts.append(tab[:N_PREAMBLE_ROWS * SYM_LEN])
        else:
            parts.append(t.fft_preamble if user is None else pnc_preamble(user))
    parts.append(scaled.reshape(-1))
    if zerogap:
        row = np.zeros(SYM_LEN, dtype=np.complex128) if gap_row is None else \
            np.asarray(gap_row).astype(np.complex64).astype(np.complex128)[:SYM_LEN]
        parts.append(np.tile(row, N_GAP_ROWS))
    # the amplitude block holds its constant as a complex64 (gr.multiply_const_cc)
    s = np.concatenate(parts) * float(np.float32(amp))
    if swap_iq:
        s = s.imag + 1j * s.real
    return {"mapped": mapped, "flags": flags, "pilot": pil, "cmap": bins, "ifft": x,
            "cp": cp, "scale": scaled, "iq": s}


def tx_packet(payload, regime, symboltime=4.0, pilot_offset=0, amp=0.3, swap_iq=False,
              preamble=True, zerogap=True, role=None):
    """Whole chain for one payload: send_pkt (ftw_ofdm.py:191-219) + flowgraph + top-level amp."""
    enc = encode_bits(payload, regime, symboltime)
    mod = modulate(enc["interleaved"], regime, pilot_offset, amp, swap_iq, preamble, zerogap, role)
    enc.update(mod)
    return enc


def tx_stream(payloads, regime, symboltime=4.0, amp=0.3, swap_iq=False):
    """A sequence of packets through one flowgraph: the pilot block's counter keeps running
    across packets (ftw_ofdm_pilot_cc.cc:82,159)."""
    off = 0
    out = []
    for p in payloads:
        r = tx_packet(p, regime, symboltime, off, amp, swap_iq)
        off += r["info"]["N_sym"] + 1
        out.append(r["iq"])
    return np.concatenate(out)


def n_samples(n_sym, preamble=True, zerogap=True):
    return SYM_LEN * ((N_PREAMBLE_ROWS if preamble else 0) + 1 + n_sym + (N_GAP_ROWS if zerogap else 0))
