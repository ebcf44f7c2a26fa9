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
