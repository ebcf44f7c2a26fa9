"""CPU restatement (numpy) of the rawofdm transmit chain: carrier/preamble set-up, mapper,
preamble insertion, shifted N-point inverse FFT, cyclic prefix, 1/sqrt(N).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else); the product never calls it.

PARITY UNPINNED: the reference holds no golden vector for this chain (its qa_ofdm.py only
asserts TX power ~ 1 and a loop-back SNR > 40 dB, rawofdm/src/python/qa_ofdm.py:65-67); the
IFFT / cyclic prefixer / preamble inserter are GNU Radio blocks whose source is not part of the
reference tree (gnuradio-core gr.fft_vcc, gr-digital ofdm_cyclic_prefixer and
ofdm_insert_preamble, version unpinned), restated here from their documented behaviour.

Citations are relative to the reference root (rawofdm/...).
"""
import cmath
import math
import random

import numpy as np


class Options(object):
    """Stand-in for the parsed option object (rawofdm/src/python/raw_ofdm_params.py:104-119)."""

    def __init__(self, fft_length=64, occupied_tones=52, cp_length=16, num_preambles=2, half_sync=True,
                 verbose=False, log=False, size=16):
        self.fft_length = fft_length
        self.occupied_tones = occupied_tones
        self.cp_length = cp_length
        self.num_preambles = num_preambles
        self.half_sync = half_sync
        self.verbose = verbose
        self.log = log
        self.size = size


class OfdmParams(object):
    """rawofdm/src/python/raw_ofdm_params.py:13-99 (Python 2 integer `/` written as `//`)."""

    def pad_symbol(self, symbol):                                   # :24-27
        padded = self.fft_length * [0, ]
        padded[self.pad: self.pad + self.occupied_tones + 1] = symbol
        return padded

    def random_symbol(self):                                        # :29-37
        psk = lambda: cmath.exp(1j * 2 * math.pi * random.random())
        symbol = [psk() for _ in range(self.occupied_tones)]
        half = self.occupied_tones // 2
        return symbol[:half] + [0] + symbol[half:]

    def make_even(self, symbol):                                    # :39-48
        sf = math.sqrt(2)
        for i in range(len(symbol)):
            if (self.pad + i) & 1:
                symbol[i] = 0
            else:
                symbol[i] *= sf

    def __init__(self, options):                                    # :50-99
        self.fft_length = int(math.ceil(options.fft_length / 2.0) * 2)
        self.occupied_tones = int(math.ceil(options.occupied_tones / 2.0) * 2)
        self.cp_length = options.cp_length
        self.num_preambles = options.num_preambles
        pad = (self.fft_length - self.occupied_tones) // 2
        self.pad = pad
        random.seed(9817)
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
        for i in range(5, half, 13):
            carriers[i] = 2
            num_pilots += 1
        carriers.extend([0] + list(reversed(carriers)))
        self.carriers = carriers
        padded_carriers = self.fft_length * [0, ]
        padded_carriers[pad: pad + self.occupied_tones + 1] = carriers
        self.padded_carriers = padded_carriers
        self.pilot_tones = num_pilots * 2
        self.data_tones = self.occupied_tones - self.pilot_tones
        self.occupied_tones += 1


def mapper(carrier_map, data, enable=None):
    """rawofdm/src/lib/ofdm/raw_ofdm_mapper.cc:22-80.  data: (n, data_tones) complex.
    Symbols whose enable flag is 0 are not written by the reference (stale buffer); they are
    returned as zeros here."""
    cm = np.asarray(carrier_map)
    if np.any((cm < 0) | (cm > 2)):
        raise ValueError("raw_ofdm_mapper: carrier_map must include only 0,1,2")     # :37-38
    data_idx = np.flatnonzero(cm == 1)
    pilot_idx = np.flatnonzero(cm == 2)
    data = np.asarray(data, dtype=np.complex64).reshape(-1, len(data_idx))
    out = np.zeros((data.shape[0], len(cm)), dtype=np.complex64)                     # :57
    pil = np.where(np.arange(len(pilot_idx)) % 2 == 0, 1.0, -1.0)                    # :62-66
    out[:, pilot_idx] = pil
    out[:, data_idx] = data                                                          # :69-71
    if enable is not None:
        out[np.asarray(enable) == 0] = 0
    return out


def insert_preamble(symbols, flags, padded_preambles):
    """gr-digital ofdm_insert_preamble as wired at rawofdm/src/python/raw_ofdm_mod.py:45,50-52:
    symbols before the first set flag are dropped; a set flag emits every preamble symbol and
    then the flagged symbol; following symbols are copied until the next flag."""
    pre = np.asarray(padded_preambles, dtype=np.complex64)
    out = []
    started = False
    for s, f in zip(symbols, flags):
        if f & 1:
            started = True
            out.extend(pre)
        if started:
            out.append(s)
    n = symbols.shape[1]
    return np.asarray(out, dtype=np.complex64).reshape(-1, n)


def ifft_cp_scale(freq_symbols, cp_length):
    """gr.fft_vcc(N, False, [], True) -> ofdm_cyclic_prefixer(N, N+cp) -> multiply_const(1/sqrt(N))
    (rawofdm/src/python/raw_ofdm_mod.py:46-48,52).  The `shift` flag makes the block read its
    input with DC in the middle: bin k of the transform is input[(k + N/2) mod N]; the backward
    transform is unnormalised.  float64 arithmetic; the product is checked against this within a
    relative L2 tolerance."""
    x = np.asarray(freq_symbols, dtype=np.complex128)
    n = x.shape[1]
    t = np.fft.ifft(np.roll(x, -(n // 2), axis=1), axis=1) * n
    t = np.concatenate([t[:, n - cp_length:], t], axis=1)
    return t * float(np.float32(1.0 / math.sqrt(n)))


def modulate(params, data, flags):
    """ofdm_mod (rawofdm/src/python/raw_ofdm_mod.py:20-52): (n, data_tones) complex + one flag per
    symbol -> rows of fft_length + cp_length samples (complex128)."""
    mapped = mapper(params.padded_carriers, data)
    with_pre = insert_preamble(mapped, flags, params.padded_preambles)
    return ifft_cp_scale(with_pre, params.cp_length)


def frame_flags(n_frames, size):
    """rawofdm/examples/ofdm_rxtx.py:81-87: one set flag followed by size-1 zeros per frame."""
    f = np.zeros((n_frames, size), dtype=np.uint8)
    f[:, 0] = 1
    return f.reshape(-1)


# =====================================================================================
# TX bit chain in front of the mapper: pad -> conv_enc -> conv_punc -> intrlv_bit -> qam_enc
# (rawofdm/src/python/raw_qam.py:76-102), plus the receive halves needed to state the
# reference's own round-trip tests (qa_intrlv.py, qa_qam.py, qa_conv.py, qa_qamconv.py).
# PARITY UNPINNED: those tests are the only checks the reference holds for this chain.
# =====================================================================================

# rawofdm/src/python/raw_qam.py:19-28: bitrate -> (bits per carrier, (nc, np))
BITRATES = {1: (1, (4, 4)), 2: (1, (6, 4)), 3: (2, (4, 4)), 4: (2, (6, 4)),
            5: (4, (4, 4)), 6: (4, (6, 4)), 7: (6, (4, 3)), 8: (6, (6, 4))}

# rawofdm/src/lib/spiral/config.h selects CODE 121-91 unless the build overrides it:
# K = 7, POLYS {121, 91} (decimal) -- rawofdm/src/lib/spiral/defs.h:3-7
CONV_K = 7
CONV_POLYS = (121, 91)


def framebytes(bitrate, size):
    """rawofdm/src/python/raw_qam.py:30-37; size = complex values per frame."""
    qambits, (nc, np_) = bitrate
    np_ *= 2
    assert (size * qambits * nc) % (np_ * 8) == 0
    return size * qambits * nc // (np_ * 8)


def conv_enc(data, sr=0):
    """rawofdm/src/lib/qam/raw_conv.cc:33-55: bytes MSB first -> 2 output items (0/1) per bit;
    the newest bit enters the register at its LSB.  Returns (items, register)."""
    out = []
    for b in bytes(data):
        for j in range(8):
            bit = (b >> (7 - j)) & 1
            sr = ((sr << 1) | bit) & 0xFFFFFFFF
            for poly in CONV_POLYS:
                out.append(bin(sr & poly).count("1") & 1)
    return np.array(out, dtype=np.uint8), sr


def conv_punc(items, nc, np_, zero=0):
    """rawofdm/src/lib/qam/raw_conv.cc:141-163: per group of nc input items np_ output items:
    the first min(nc, np_) copied, the rest `zero`."""
    items = np.asarray(items, dtype=np.uint8)
    n_groups = len(items) // nc
    src = items[:n_groups * nc].reshape(n_groups, nc)
    out = np.full((n_groups, np_), zero, dtype=np.uint8)
    m = min(nc, np_)
    out[:, :m] = src[:, :m]
    return out.reshape(-1)


def intrlv_bit_map(ncarriers, nbits):
    """rawofdm/src/lib/qam/raw_intrlv.cc:14-31: destination of item k inside one symbol."""
    cbps = nbits * ncarriers
    k = np.arange(cbps)
    s = max(nbits // 2, 1)
    i = (cbps // 16) * (k % 16) + k // 16
    return s * (i // s) + (i + cbps - (16 * i // cbps)) % s


def intrlv_bit(items, ncarriers, nbits, inverse):
    """rawofdm/src/lib/qam/raw_intrlv.cc:32-76: out[d_map[j]] = in[j] per symbol; the inverse
    block uses the inverted map."""
    m = intrlv_bit_map(ncarriers, nbits)
    if inverse:
        inv = np.empty_like(m)
        inv[m] = np.arange(len(m))
        m = inv
    items = np.asarray(items)
    body = items[:len(items) // len(m) * len(m)].reshape(-1, len(m))
    out = np.empty_like(body)
    out[:, m] = body
    return out.reshape(-1)


def qam_scale(numbits, power):
    """rawofdm/src/lib/qam/raw_qam.cc:39-54, float32 arithmetic as written there."""
    nn = 1 << (numbits - 1)
    sum2 = (4 * nn * nn * nn - nn) // 3
    return np.float32(np.sqrt(np.float64(np.float32(power) * np.float32(nn) / np.float32(sum2))))


_QAM = {1: (1, 1.0), 2: (1, 0.5), 4: (2, 0.5), 6: (3, 0.5)}       # raw_qam.cc:120-147: bits per axis, power


def qam_axis(bits, numbits, sf):
    """raw_qam.cc:63-74: recursive Gray coding of numbits items into one PAM level."""
    bits = np.asarray(bits, dtype=np.int64).reshape(-1, numbits)
    pt = np.zeros(len(bits), dtype=np.int64)
    flip = np.ones(len(bits), dtype=np.int64)
    for i in range(numbits):
        bit = bits[:, i] * 2 - 1
        pt = bit * flip + pt * 2
        flip = flip * -bit
    return pt.astype(np.float32) * sf


def qam_enc(items, nbits):
    """raw_qam.cc:113-150.  BPSK writes the real part only (the imaginary float is left
    untouched by the reference; zero here)."""
    numbits, power = _QAM[nbits]
    sf = qam_scale(numbits, power)
    items = np.asarray(items, dtype=np.uint8)
    n = len(items) // nbits
    if nbits == 1:
        re = qam_axis(items[:n], 1, sf)
        return (re + 0j).astype(np.complex64)
    lv = qam_axis(items[:n * nbits], numbits, sf).reshape(n, 2)
    return (lv[:, 0] + 1j * lv[:, 1]).astype(np.complex64)


def qam_dec(sym, nbits, gain=0):
    """raw_qam.cc:83-96,166-203: per-bit confidences 0..255 (128 = undecided)."""
    numbits, power = _QAM[nbits]
    sf = qam_scale(numbits, power)
    d_gain = gain + 8 - numbits
    scale_d = np.float32((1 << d_gain) / sf)
    sym = np.asarray(sym, dtype=np.complex64)
    axes = sym.real[:, None] if nbits == 1 else np.stack([sym.real, sym.imag], axis=1)
    pt = (axes.astype(np.float32) * scale_d).astype(np.int64)       # C float -> int truncation
    flip = np.ones_like(pt)
    amp = (1 << (numbits - 1)) << d_gain
    out = []
    for _ in range(numbits):
        out.append(np.clip(flip * pt + 128, 0, 255))
        bit = np.where(pt < 0, -1, 1)
        pt = pt - bit * amp
        flip = -bit
        amp //= 2
    return np.stack(out, axis=2).reshape(-1).astype(np.uint8)


def viterbi_decode(soft, n_bits):
    """Maximum-likelihood decoder for conv_enc (stands in for the Spiral decoder behind
    raw_conv_dec, raw_conv.cc:87-104): soft items 0..255 (128 = erasure), start state 0, the last
    K-1 data bits are zero padding.  Returns n_bits/8 bytes (the padding byte included)."""
    soft = np.asarray(soft, dtype=np.int64).reshape(-1, 2)[:n_bits]
    n_states = 1 << (CONV_K - 1)
    states = np.arange(n_states)
    # state = the previous K-1 bits, newest at the LSB; next register = (state << 1 | bit)
    par = lambda v: np.array([bin(int(x)).count("1") & 1 for x in v])
    metric = np.full(n_states, 1 << 40, dtype=np.int64)
    metric[0] = 0
    decisions = np.zeros((len(soft), n_states), dtype=np.uint8)
    prev0 = states >> 1                                   # predecessor with oldest bit 0
    prev1 = prev0 | (n_states >> 1)                       # ... with oldest bit 1
    bit_in = states & 1
    exp = {}
    for name, pv in (("p0", prev0), ("p1", prev1)):
        reg = (pv << 1) | bit_in
        exp[name] = np.stack([par(reg & CONV_POLYS[0]), par(reg & CONV_POLYS[1])], axis=1)
    for t, (s0, s1) in enumerate(soft):
        cost = lambda e: np.where(e[:, 0] == 1, 255 - s0, s0) + np.where(e[:, 1] == 1, 255 - s1, s1)
        m0 = metric[prev0] + cost(exp["p0"])
        m1 = metric[prev1] + cost(exp["p1"])
        decisions[t] = m1 < m0
        metric = np.minimum(m0, m1)
    state = 0                                             # zero padding drives the encoder back to 0
    bits = np.zeros(len(soft), dtype=np.uint8)
    for t in range(len(soft) - 1, -1, -1):
        bits[t] = state & 1
        state = (state >> 1) | (int(decisions[t, state]) << (CONV_K - 2))
    return np.packbits(bits).tobytes()


def qam_tx(bitrate, ncarriers, nsymbols, data):
    """rawofdm/src/python/raw_qam.py:76-102: whole frames of `framebytes` bytes -> nsymbols *
    ncarriers complex values per frame.  Returns (values, taps)."""
    qambits, (nc, np_) = BITRATES[bitrate]
    fb = framebytes((qambits, (nc, np_)), ncarriers * nsymbols) - 1
    data = bytes(data)
    assert len(data) % fb == 0
    padded = b"".join(data[i:i + fb] + b"\x00" for i in range(0, len(data), fb))     # conv_punc(fb, fb + 1)
    coded, _ = conv_enc(padded)
    punct = conv_punc(coded, nc, np_)
    itl = intrlv_bit(punct, ncarriers, qambits, False)
    return qam_enc(itl, qambits), {"pad": padded, "conv": coded, "punc": punct, "intrlv": itl, "framebytes": fb}


def qam_rx(bitrate, ncarriers, nsymbols, values):
    """rawofdm/src/python/raw_qam.py:39-64 with viterbi_decode in place of the Spiral decoder."""
    qambits, (nc, np_) = BITRATES[bitrate]
    fb = framebytes((qambits, (nc, np_)), ncarriers * nsymbols) - 1
    soft = qam_dec(values, qambits)
    de = intrlv_bit(soft, ncarriers, qambits, True)
    dep = conv_punc(de, np_, nc, 128)
    per_frame = (fb + 1) * 16
    out = b""
    for i in range(0, len(dep), per_frame):
        out += viterbi_decode(dep[i:i + per_frame], (fb + 1) * 8)[:fb]                # depad
    return out
