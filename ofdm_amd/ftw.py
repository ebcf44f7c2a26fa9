"""Factories with the names and signatures of the reference's SWIG module `gnuradio.ftw`
(ftw_80211_ofdm_tx/src/ftw/ftw.i:42-218).  The objects carry the block parameters and check
them the way the C++ constructors / work functions do; the arithmetic of all of them runs fused
in the HIP kernels (ofdmtx_modulate / ofdmtx_tx), driven by ofdm_amd.ftw_ofdm.ofdm_mod, which
hands every held parameter to the library: the mapper's constellation (ofdmtx_set_constellation),
the preamble table (ofdmtx_set_preamble) and the gap row (ofdmtx_set_gap).
"""
from . import _lib
from . import gr


def ftw_crc32(s):
    """ftw.i:215-218, ftw_crc32.cc:72-82."""
    if isinstance(s, str):
        s = s.encode("latin-1")
    return _lib.crc32(s)


def ftw_update_crc32(crc, s):
    """ftw.i:215-218, ftw_crc32.cc:33-64: the bit-reversed CRC register after `s`, without the final
    inversion; `crc` is accepted and ignored as there (ftw_crc32.cc:53 restarts from all ones)."""
    if isinstance(s, str):
        s = s.encode("latin-1")
    return _lib.update_crc32(crc, s)


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


class pnc_ofdm_pilot_cc(object):
    """ftw.i:66-68, ftw_pnc_ofdm_pilot_cc.cc:125-131,146-178: constant pilots at -21, -7, +7, +21 --
    user 1: (1, 0, 1, 0); any other non-zero user: (0, 1, 0, 1); user 0: (1, 1, 1, 1).  No
    polarity sequence and no sign change on the fourth pilot."""

    def __init__(self, tones, user):
        self.tones, self.user = tones, int(user)

    def check(self):
        if self.tones != 48:                                        # :121-122
            raise RuntimeError("ftw_pnc_ofdm_pilot_cc: IEEE 802.11p requires 48 subcarriers\n")

    @property
    def pilot_mode(self):
        return {1: _lib.PILOT_PNC_A, 0: _lib.PILOT_PNC_ALL}.get(self.user, _lib.PILOT_PNC_B)


class ofdm_cmap_cc(object):
    """ftw.i:42-50; ftw_ofdm_cmap_cc.cc:97-99."""

    def __init__(self, fft_size, tones):
        self.fft_size, self.tones = fft_size, tones

    def check(self):
        if self.fft_size != 64:
            raise RuntimeError("ftw_ofdm_cmap_cc: IEEE 802.11p requires 64 FFT size\n")


def _rows_for_80(symbol_length, what):
    # the fused kernel writes rows of 80 samples (64-point transform + 16 samples of cyclic prefix)
    if int(symbol_length) != 80:
        raise RuntimeError("%s: symbol_length must be 80 (%s)" % (what, _lib.load().ofdmtx_strerror(_lib.ESHAPE).decode()))


class ofdm_preamble(object):
    """ftw.i:78-90, ftw_ofdm_preamble.cc:39-99: the first four rows of symbol_length samples of
    preamble[0] in front of every packet (:79-83)."""

    def __init__(self, symbol_length, N_symbols, preamble):
        _rows_for_80(symbol_length, "ftw_ofdm_preamble")
        self.symbol_length, self.N_symbols = symbol_length, N_symbols
        self.preamble = [list(p) for p in preamble]
        if not self.preamble or len(self.preamble[0]) < 4 * symbol_length:
            # the reference would copy from beyond the table
            raise ValueError("ftw_ofdm_preamble: preamble[0] must hold 4 x %d samples" % symbol_length)


class pnc_ofdm_preamble(object):
    """ftw.i:94-106, ftw_pnc_ofdm_preamble.cc:80-107: six rows in front of every packet -- the two
    short-training rows of the table, then its two long-training rows followed (user 1) or
    preceded (user 2) by two rows of zeros."""

    def __init__(self, symbol_length, N_symbols, preamble, user):
        if int(user) not in (1, 2):                                 # :105 only prints an error
            raise RuntimeError("ERROR: user not set properly! [FTW_PNC_PREAMBLE]")
        _rows_for_80(symbol_length, "ftw_pnc_ofdm_preamble")
        self.symbol_length, self.N_symbols, self.user = symbol_length, N_symbols, int(user)
        self.preamble = [list(p) for p in preamble]
        if not self.preamble or len(self.preamble[0]) < 4 * symbol_length:
            raise ValueError("ftw_pnc_ofdm_preamble: preamble[0] must hold 4 x %d samples" % symbol_length)

    @property
    def flags(self):
        return _lib.F_PREAMBLE | (_lib.F_PNC_A if self.user == 1 else _lib.F_PNC_B)


class zerogap(object):
    """ftw.i:110-122, ftw_zerogap.cc:39-102: gap[0] (one row of symbol_length samples; the
    reference passes zeros) 13 times behind every packet (:88-92)."""

    def __init__(self, symbol_length, N_symbols, gap):
        _rows_for_80(symbol_length, "ftw_zerogap")
        self.symbol_length, self.N_symbols = symbol_length, N_symbols
        self.gap = [list(g) for g in gap]
        if not self.gap or len(self.gap[0]) < symbol_length:
            raise ValueError("ftw_zerogap: gap[0] must hold %d samples" % symbol_length)


class pnc_zerogap(zerogap):
    """ftw.i:126-138, ftw_pnc_zerogap.cc:39-102: passes N_symbols + 7 rows (six preamble rows and
    the SIGNAL symbol more than DATA symbols), then the same 13 rows of zeros."""


class repetition(object):
    """ftw.i:143-146, ftw_repetition.cc:40-76: emits the first packet of its input -- (N_symbols +
    5 + 13) rows of symbol_length samples -- over and over (the block never consumes input; its
    `repetition` argument is unused, :72-74 is commented out)."""

    def __init__(self, symbol_length, repetition, N_symbols):
        self.symbol_length, self.repetition_count, self.N_symbols = int(symbol_length), repetition, int(N_symbols)

    @property
    def period(self):
        return (self.N_symbols + 5 + 13) * self.symbol_length

    def work(self, in_samples, noutput_items, ctx=None, stream=None):
        """in_samples: complex64 device tensor holding at least one period; returns noutput_items
        rows (noutput_items * symbol_length samples)."""
        import torch
        from . import ftw_packet_utils
        if in_samples.dtype != torch.complex64 or not in_samples.is_contiguous():
            raise TypeError("repetition: contiguous complex64 tensor expected")
        if in_samples.numel() < self.period:
            raise ValueError("repetition: input shorter than one packet (%d samples)" % self.period)
        ctx = ctx or ftw_packet_utils._context(in_samples.device.index)
        n_out = int(noutput_items) * self.symbol_length
        out = torch.empty(n_out, dtype=torch.complex64, device=in_samples.device)
        st = stream if stream is not None else torch.cuda.current_stream(in_samples.device).cuda_stream
        _lib.check(_lib.load().ofdmtx_repeat(ctx._h, in_samples.data_ptr(), self.period, n_out,
                                             out.data_ptr(), st))
        return out
