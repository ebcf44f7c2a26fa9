"""Drop-in for the reference's ftw_80211_ofdm_tx/src/examples/ftw_ofdm.py: `ofdm_mod` and
`ftw_transmit_path` with the same constructor arguments and the same call sequence
(ofdm_mod(options, payload) -> send_pkt(payload) x N -> send_pkt(eof=True)).

Where the reference runs per-bit Python per packet and a thread-per-block flowgraph, send_pkt
here only queues the payload; queued packets are modulated as ONE batch by the fused HIP kernel
(ofdmtx_tx) when the batch is full or the stream ends, and the IQ is handed to the connected sinks.
The pilot polarity counter runs on over all packets of the process, as the reference's global
counter does (ftw/ftw_ofdm_pilot_cc.cc:82,159).
"""
import collections
import math

import numpy as np
import torch

from . import _lib
from . import ftw
from . import gr
from . import ftw_packet_utils as ofdm_packet_utils
from .batch import StreamOutput, TxContext

# symbols modulated so far in this process (ftw/ftw_ofdm_pilot_cc.cc:82: a file-scope counter)
_pilot_offset = [0]


class ofdm_mod(gr.hier_block2):
    """examples/ftw_ofdm.py:45-219.

    options: regime (str "1".."8"), interp, swapIQ, log, role (None, 'A' or 'B': the PNC pilot,
    preamble and zero-gap blocks of :101-106,133-138,147-149).  batch_packets bounds
    how many queued packets are modulated per kernel launch.  spread_output: see batch.StreamOutput
    (where the IQ of large batches is written)."""

    def __init__(self, options, payload='', msgq_limit=2, pad_for_usrp=False, batch_packets=65536, device=None,
                 spread_output=None):
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
        # This block chain owns its library context: the mapper's constellation, the preamble table
        # and the gap row are per-instance block arguments in the reference (ftw/ftw.i:78-122,157-163).
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self._ctx = TxContext(device)
        self._ctx.set_constellation(self._pkt_input.nbits, rotated_const)
        self._ctx.set_preamble(self.preamble.preamble[0])            # ftw/ftw_ofdm_preamble.cc:79-83 reads row 0
        self._ctx.set_gap(self.zerogap.gap[0])                       # ftw/ftw_zerogap.cc:88-92 reads row 0
        self._batch_packets = int(batch_packets)
        self._pending = []
        self._outs = []              # (multiply_const_cc or None, sink) per connection
        self._eof = False
        self._log_files = {}
        self._plans = collections.OrderedDict()    # uniform plans of recent batches
        self._live = []              # plans and buffers of the last flush: kept until the next one has been enqueued
        self._output = StreamOutput(self._ctx, spread_output)     # ring of output slots + hand-over to the sinks
        self._pinned = None          # page-locked staging buffer for the payload bytes
        self._pinned_free = None     # event: the copy out of it has finished
        self._dev_payloads = None

    # ---- wiring (ofdm_amd.gr.top_block.connect)
    def _attach(self, gain, sinks):
        self._outs.extend((gain, s) for s in sinks)

    def _start(self):
        pass

    def _finish(self):
        self._flush()
        self._output.drain()
        for f in self._log_files.values():
            f.close()
        self._log_files = {}

    # ---- the reference's interface
    def send_pkt(self, payload='', eof=False):
        """examples/ftw_ofdm.py:191-219."""
        if eof:
            self._eof = True
            self._flush()
            self._output.drain()
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

    def _stage_payloads(self, pk, total):
        """Payload bytes -> device through a page-locked buffer that is kept between flushes (a
        pageable source makes the runtime stage every copy itself, at a fraction of the link rate)."""
        dev = self._ctx.device
        need = max(total, 1)
        if self._pinned is None or self._pinned.numel() < need:
            cap = max(need, 1 << 20, 2 * (self._pinned.numel() if self._pinned is not None else 0))
            if self._pinned_free is not None:
                self._pinned_free.synchronize()
            self._pinned = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
            self._dev_payloads = torch.empty(cap, dtype=torch.uint8, device=dev)
        elif self._pinned_free is not None:
            self._pinned_free.synchronize()                     # the previous copy still reads the buffer
        # one join and one copy (a Python loop over 10^5 payloads costs more than the transfer)
        self._pinned.numpy()[:total] = np.frombuffer(b"".join(pk), dtype=np.uint8)
        out = self._dev_payloads[:need]
        out.copy_(self._pinned[:need], non_blocking=True)
        self._pinned_free = torch.cuda.Event()
        self._pinned_free.record(torch.cuda.current_stream(dev))
        return out

    def _plan(self, lens, amp, flags, pilot_mode, start):
        ctx = self._ctx
        if np.all(lens == lens[0]):
            key = (len(lens), int(lens[0]), float(amp), int(flags), int(pilot_mode), int(start))
            plan = self._plans.get(key)
            if plan is None:
                plan = ctx.plan_uniform(len(lens), int(self._regime), int(lens[0]), symbol_time=self._symbol_time,
                                        pilot_mode=pilot_mode, pilot_start=start, flags=flags, amp=amp)
                self._plans[key] = plan
                if len(self._plans) > 256:
                    self._plans.popitem(last=False)
            return plan
        return ctx.plan_ragged(np.full(len(lens), int(self._regime), np.uint8), lens,
                               symbol_time=self._symbol_time, pilot_mode=pilot_mode, pilot_start=start,
                               flags=flags, amp=amp)

    def _flush(self):
        if not self._pending:
            return
        pk, self._pending = self._pending, []
        framing = getattr(self.preamble, "flags", _lib.F_PREAMBLE) | _lib.F_ZEROGAP
        flags = framing | (_lib.F_SWAPIQ if self._swap else 0)
        pilot_mode = getattr(self.pilot, "pilot_mode", _lib.PILOT_REF_STREAM)
        lens = np.fromiter((len(p) for p in pk), dtype=np.uint32, count=len(pk))
        start = _pilot_offset[0] % 127
        payloads = self._stage_payloads(pk, int(lens.sum()))
        live = [payloads]
        unit = None                  # the path's own output: amplitude 1, no I/Q exchange
        if self._log:
            # examples/ftw_ofdm.py:181-189: one file per block, named and shaped as there; all taps sit
            # in front of tx_amplitude and of the I/Q exchange.  The bit-level stages of send_pkt are added.
            p1 = self._plan(lens, 1.0, framing, pilot_mode, start)
            bits = p1.encode_bits(payloads, taps=True)
            unit, front = p1.modulate(bits["interleaved"], taps=True)
            back = p1.symbol_taps(unit)
            self._tap("ofdm_mapper.dat", front["mapped"])
            self._tap("ofdm_pilot.dat", front["pilot"])
            self._tap("ofdm_cmap.dat", front["cmap"])
            self._tap("ofdm_ifft.dat", back["ifft"])
            self._tap("ofdm_cp_adder.dat", back["cp"])
            self._tap("ofdm_scale.dat", back["scale"])
            self._tap("ofdm_preamble.dat", back["preamble"])
            self._tap("ofdm_zerogap.dat", unit)
            for k in ("plcp", "scrambled", "coded", "interleaved"):
                self._tap("ofdm_%s.dat" % k, bits[k])
            live.append(p1)
        # one kernel launch per distinct amplitude among the connections (the amplitude is folded
        # into the transform's twiddles, examples/ftw_ofdm_tx_uhd.py:82-86)
        # Large batches are written into the slots of a ring whose memory alternates between the parts of
        # the device (StreamOutput) and reach the sinks one batch late, under the next batch's kernels.
        outs = self._outs or [(None, None)]
        amps = {float(g.k()) if g is not None else 1.0 for g, _ in outs}
        done, writes = {}, []
        n_symbols, ringed = None, False
        for gain, sink in outs:
            amp = float(gain.k()) if gain is not None else 1.0
            if amp not in done:
                if amp == 1.0 and unit is not None and not self._swap:
                    done[amp] = unit
                else:
                    plan = self._plan(lens, amp, flags, pilot_mode, start)
                    slot = self._output.buffer(plan.n_samples, len(amps))
                    ringed = ringed or slot is not None
                    done[amp] = plan.tx(payloads, out=slot)[:plan.n_samples]
                    n_symbols = plan.n_symbols
                    live.append(plan)
            if sink is not None:
                writes.append((sink, done[amp]))
        if n_symbols is None:
            n_symbols = live[-1].n_symbols
        _pilot_offset[0] += n_symbols
        # (a ring slot is written again two batches later: last_iq is good until then)
        self.last_iq = next(iter(done.values()))
        self._output.hand_over(writes, ringed)
        self._live = live            # the previous flush's plans and buffers may go now: this one is enqueued behind them


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
