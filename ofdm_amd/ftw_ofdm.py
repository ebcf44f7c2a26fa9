"""Drop-in for the reference's ftw_80211_ofdm_tx/src/examples/ftw_ofdm.py: `ofdm_mod` and
`ftw_transmit_path` with the same constructor arguments and the same call sequence
(ofdm_mod(options, payload) -> send_pkt(payload) x N -> send_pkt(eof=True)).

Where the reference runs per-bit Python per packet and a thread-per-block flowgraph, send_pkt
here only queues the payload; queued packets are modulated as ONE batch by the fused HIP kernel
(ofdmtx_tx) when the batch is full or the stream ends, and the IQ is handed to the connected sinks.
The pilot polarity counter runs on over all packets of the process, as the reference's global
counter does (ftw/ftw_ofdm_pilot_cc.cc:82,159).
"""
import math

import numpy as np
import torch

from . import _lib
from . import ftw
from . import gr
from . import ftw_packet_utils as ofdm_packet_utils

# symbols modulated so far in this process (ftw/ftw_ofdm_pilot_cc.cc:82: a file-scope counter)
_pilot_offset = [0]


class ofdm_mod(gr.hier_block2):
    """examples/ftw_ofdm.py:45-219.

    options: regime (str "1".."8"), interp, swapIQ, log, role (None, 'A' or 'B': the PNC pilot,
    preamble and zero-gap blocks of :101-106,133-138,147-149).  batch_packets bounds
    how many queued packets are modulated per kernel launch."""

    def __init__(self, options, payload='', msgq_limit=2, pad_for_usrp=False, batch_packets=65536, device=None):
        gr.hier_block2.__init__(self, "ofdm_mod")
        self._fft_length = 64
        self._total_sub_carriers = 53
        self._data_subcarriers = 48
        self._cp_length = 16
        self._regime = str(options.regime)
        self._symbol_length = self._fft_length + self._cp_length
        self._role = getattr(options, "role", None)
        # examples/ftw_ofdm.py:81 (Python 2 `/`: interp is a float there, ftw_ofdm_tx_uhd.py:71)
        self._symbol_time = options.interp * (self._fft_length + self._cp_length) / 100

        consts = {"1": ofdm_packet_utils.bpsk, "2": ofdm_packet_utils.bpsk, "3": ofdm_packet_utils.qpsk,
                  "4": ofdm_packet_utils.qpsk, "5": ofdm_packet_utils.qam16, "6": ofdm_packet_utils.qam16,
                  "7": ofdm_packet_utils.qam64, "8": ofdm_packet_utils.qam64}
        if self._regime not in consts:
            raise UnboundLocalError("local variable 'rotated_const' referenced before assignment")
        rotated_const = consts[self._regime](self)

        # the blocks of examples/ftw_ofdm.py:98-149: parameter holders, checked as the C++ blocks check
        self._pkt_input = ftw.ofdm_mapper(rotated_const, msgq_limit, self._data_subcarriers, self._fft_length)
        if self._role == 'A':                                        # examples/ftw_ofdm.py:101-109
            self.pilot = ftw.pnc_ofdm_pilot_cc(self._data_subcarriers, 1)
        elif self._role == 'B':
            self.pilot = ftw.pnc_ofdm_pilot_cc(self._data_subcarriers, 2)
        else:
            self.pilot = ftw.ofdm_pilot_cc(self._data_subcarriers)
        self.cmap = ftw.ofdm_cmap_cc(self._fft_length, self._total_sub_carriers)
        self.pilot.check()
        self.cmap.check()
        self.scale = gr.multiply_const_cc(1.0 / math.sqrt(self._fft_length))
        info = ofdm_packet_utils.get_info(payload, options.regime, self._symbol_time)
        N_sym = info["N_sym"]
        if self._role in ('A', 'B'):                                 # examples/ftw_ofdm.py:133-149
            self.preamble = ofdm_packet_utils.insert_preamble(self._symbol_length, N_sym, self._role)
        else:
            self.preamble = ofdm_packet_utils.insert_preamble(self._symbol_length, N_sym)
        if self._role is None:
            self.zerogap = ofdm_packet_utils.insert_zerogap(self._symbol_length, N_sym)
        else:
            self.zerogap = ofdm_packet_utils.insert_zerogap(self._symbol_length, N_sym, 'FPNC')

        self._swap = bool(getattr(options, "swapIQ", False))
        self._log = bool(getattr(options, "log", False))
        self._ctx = ofdm_packet_utils._context(device)
        self._ctx.set_constellation(self._pkt_input.nbits, rotated_const)
        self._batch_packets = int(batch_packets)
        self._pending = []
        self._gain = None
        self._sinks = []
        self._eof = False
        self._log_files = {}

    # ---- wiring (ofdm_amd.gr.top_block.connect)
    def _attach(self, gain, sinks):
        if gain is not None:
            self._gain = gain
        self._sinks.extend(sinks)

    def _start(self):
        pass

    def _finish(self):
        self._flush()
        for f in self._log_files.values():
            f.close()
        self._log_files = {}

    # ---- the reference's interface
    def send_pkt(self, payload='', eof=False):
        """examples/ftw_ofdm.py:191-219."""
        if eof:
            self._eof = True
            self._flush()
            return
        payload = ofdm_packet_utils._as_bytes(payload)
        if 24 + len(payload) + 4 > ofdm_packet_utils.MAXLEN:
            raise ValueError("MPDU-length must be in [0, %d]" % (ofdm_packet_utils.MAXLEN,))
        self._pending.append(payload)
        if len(self._pending) >= self._batch_packets:
            self._flush()

    def _tap(self, name, t):
        if name not in self._log_files:
            self._log_files[name] = open(name, "wb")
        t.cpu().numpy().tofile(self._log_files[name])

    def _flush(self):
        if not self._pending:
            return
        pk, self._pending = self._pending, []
        ctx = self._ctx
        amp = float(self._gain.k()) if self._gain is not None else 1.0
        flags = (getattr(self.preamble, "flags", _lib.F_PREAMBLE) | _lib.F_ZEROGAP |
                 (_lib.F_SWAPIQ if self._swap else 0))
        pilot_mode = getattr(self.pilot, "pilot_mode", _lib.PILOT_REF_STREAM)
        lens = np.fromiter((len(p) for p in pk), dtype=np.uint32, count=len(pk))
        start = _pilot_offset[0] % 127

        def make_plan(a):
            if np.all(lens == lens[0]):
                return ctx.plan_uniform(len(pk), int(self._regime), int(lens[0]), symbol_time=self._symbol_time,
                                        pilot_mode=pilot_mode, pilot_start=start, flags=flags, amp=a)
            return ctx.plan_ragged(np.full(len(pk), int(self._regime), np.uint8), lens,
                                   symbol_time=self._symbol_time, pilot_mode=pilot_mode, pilot_start=start,
                                   flags=flags, amp=a)

        plan = make_plan(amp)
        _pilot_offset[0] += plan.n_symbols
        host = torch.frombuffer(bytearray(b"".join(pk)) or bytearray(1), dtype=torch.uint8)
        payloads = host.to(ctx.device)
        if self._log:
            # examples/ftw_ofdm.py:181-189: of the per-block taps only the mapper's symbols exist as
            # data here (everything behind the mapper is fused); the bit-level stages are added
            taps = plan.encode_bits(payloads, taps=True)
            iq, mapped = plan.modulate(taps["interleaved"], want_mapped=True)
            self._tap("ofdm_mapper.dat", mapped)
            for k in ("plcp", "scrambled", "coded", "interleaved"):
                self._tap("ofdm_%s.dat" % k, taps[k])
            # output of the last block of the path, i.e. before tx_amplitude (examples/ftw_ofdm.py:189)
            self._tap("ofdm_zerogap.dat", iq if amp == 1.0 else make_plan(1.0).tx(payloads))
        else:
            iq = plan.tx(payloads)
        for s in self._sinks:
            s.write(iq)
        self.last_iq = iq


class ftw_transmit_path(gr.hier_block2):
    """examples/ftw_ofdm.py:29-42."""

    def __init__(self, options, payload=''):
        gr.hier_block2.__init__(self, "transmit_path")
        self.ofdm_tx = ofdm_mod(options, payload, msgq_limit=2, pad_for_usrp=False)

    def send_pkt(self, payload='', eof=False):
        return self.ofdm_tx.send_pkt(payload, eof)

    def _attach(self, gain, sinks):
        self.ofdm_tx._attach(gain, sinks)

    def _start(self):
        self.ofdm_tx._start()

    def _finish(self):
        self.ofdm_tx._finish()


def conv_packed_binary_string_to_1_0_string(s):
    """examples/ftw_ofdm.py:221-232."""
    return ofdm_packet_utils.conv_packed_binary_string_to_1_0_string(s)
