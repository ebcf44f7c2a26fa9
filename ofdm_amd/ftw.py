"""Factories with the names and signatures of the reference's SWIG module `gnuradio.ftw`
(ftw_80211_ofdm_tx/src/ftw/ftw.i:42-218).  The objects carry the block parameters and check
them the way the C++ constructors / work functions do; the arithmetic of all of them runs fused
in the HIP kernels (ofdmtx_modulate / ofdmtx_tx), driven by ofdm_amd.ftw_ofdm.ofdm_mod.
"""
from . import _lib
from . import gr


def ftw_crc32(s):
    """ftw.i:215-218, ftw_crc32.cc:72-82."""
    if isinstance(s, str):
        s = s.encode("latin-1")
    return _lib.crc32(s)


def ftw_update_crc32(crc, s):
    raise NotImplementedError("ftw_update_crc32 ignores its crc argument in the reference "
                              "(ftw_crc32.cc:53); use ftw_crc32")


class ofdm_mapper(object):
    """ftw.i:157-180, ftw_ofdm_mapper.cc:47-126."""

    def __init__(self, constellation, msgq_limit, occupied_carriers, fft_length):
        if not occupied_carriers <= fft_length:
            raise RuntimeError("ftw_ofdm_mapper: occupied carriers must be <= fft_length")
        if occupied_carriers != 48 or fft_length != 64:
            raise RuntimeError(_lib.load().ofdmtx_strerror(_lib.ESHAPE).decode())
        self.constellation = tuple(complex(c) for c in constellation)
        if len(self.constellation) not in (2, 4, 16, 64):
            raise RuntimeError("constellation must have 2, 4, 16 or 64 points")
        self.nbits = {2: 1, 4: 2, 16: 4, 64: 6}[len(self.constellation)]
        self.occupied_carriers = occupied_carriers
        self.fft_length = fft_length
        self._msgq = gr.msg_queue(msgq_limit)

    def msgq(self):
        return self._msgq


class ofdm_pilot_cc(object):
    """ftw.i:54-62; ftw_ofdm_pilot_cc.cc:110-111 rejects anything but 48 tones when it first runs."""

    def __init__(self, tones):
        self.tones = tones

    def check(self):
        if self.tones != 48:
            raise RuntimeError("ftw_ofdm_pilot_cc: IEEE 802.11p requires 48 subcarriers\n")


class ofdm_cmap_cc(object):
    """ftw.i:42-50; ftw_ofdm_cmap_cc.cc:97-99."""

    def __init__(self, fft_size, tones):
        self.fft_size, self.tones = fft_size, tones

    def check(self):
        if self.fft_size != 64:
            raise RuntimeError("ftw_ofdm_cmap_cc: IEEE 802.11p requires 64 FFT size\n")


class ofdm_preamble(object):
    """ftw.i:78-90, ftw_ofdm_preamble.cc:39-99: four rows of symbol_length samples in front of
    every packet."""

    def __init__(self, symbol_length, N_symbols, preamble):
        self.symbol_length, self.N_symbols = symbol_length, N_symbols
        self.preamble = [list(p) for p in preamble]


class zerogap(object):
    """ftw.i:110-122, ftw_zerogap.cc:39-102: 13 rows of zeros behind every packet."""

    def __init__(self, symbol_length, N_symbols, gap):
        self.symbol_length, self.N_symbols = symbol_length, N_symbols
        self.gap = [list(g) for g in gap]
