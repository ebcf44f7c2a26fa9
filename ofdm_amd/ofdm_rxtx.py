"""Drop-in for the transmit half of the reference's rawofdm/examples/ofdm_rxtx.py: `TX`,
`make_data` and `sender_thread` with the same constructor arguments and call sequence
(ofdm_tx.py:37-52: data = make_data(...); connect(data, tx); sender_thread(tx, options).run()).

Where the reference pushes one start-of-frame message per send_packet through a flowgraph,
send_packet here only counts frames; the frames are modulated as ONE batch by the variable-size
modulator kernel (ofdm_amd.raw.ofdm_mod.modulate_frames) when enough are queued or the stream ends.
"""
import cmath
import math
import random
import sys
import time

import numpy as np
import torch

from . import gr
from . import raw
from .batch import StreamOutput


class TX(gr.hier_block2):
    """rawofdm/examples/ofdm_rxtx.py:63-98.  Input: data_tones complex values per symbol (connect a
    vector_source_c, e.g. make_data's); output: baseband samples to the connected sinks."""

    def __init__(self, options, ctx=None, device=0, batch_frames=65536, spread_output=None):
        gr.hier_block2.__init__(self, "TX")
        self.tx = raw.ofdm_mod(options, ctx=ctx, device=device)
        self.params = self.tx.params
        self.size = options.size
        self._source = None
        self._outs = []
        self._frames = 0
        self._batch_frames = int(batch_frames)
        self.last_iq = None
        self._output = None          # ring of output slots + hand-over to the sinks (made at the first flush)
        self._spread_output = spread_output      # see batch.StreamOutput

    # ---- wiring (ofdm_amd.gr.top_block.connect)
    def _set_source(self, src):
        if getattr(src, "vlen", self.params.data_tones) != self.params.data_tones:
            raise ValueError("TX input carries %d values per symbol, the source %d" % (self.params.data_tones, src.vlen))
        self._source = src

    def _attach(self, gain, sinks):
        self._outs.extend((gain, s) for s in sinks)

    def _start(self):
        pass

    def _finish(self):
        self._flush()
        if self._output is not None:
            self._output.drain()

    # ---- the reference's interface
    def send_packet(self, num=1):
        """ofdm_rxtx.py:81-87: `num` frames of `size` symbols, each led by a set start-of-frame flag."""
        self._frames += int(num)
        if self._frames >= self._batch_frames:
            self._flush()

    def send_eof(self):
        """ofdm_rxtx.py:89-92."""
        self._flush()
        if self._output is not None:
            self._output.drain()

    def _flush(self):
        n, self._frames = self._frames, 0
        if n == 0:
            return
        if self._source is None:
            raise RuntimeError("TX: no data source connected")
        data = self._source.take(n * self.size, self.tx.ctx.device)
        # Large batches go into the slots of a ring whose memory alternates between the parts of the
        # device and reach the sinks one batch late, under the next batch's kernels (StreamOutput); there
        # a gain block scales the slot in place, once (the connections of one TX share one gain in the
        # reference's scripts, ofdm_tx.py:48-52; differing gains, and small batches, get copies).
        if self._output is None:
            self._output = StreamOutput(self.tx.ctx, self._spread_output)
        taken = []

        def where(n_samples):
            taken.append(self._output.buffer(n_samples))
            return taken[-1]

        iq = self.tx.modulate_frames(data, self.size, out=where)
        ringed = bool(taken) and taken[0] is not None
        self.last_iq = iq            # (a ring slot: scaled in place below, written again two batches later)
        ctx = self.tx.ctx
        gains = {complex(g.k()) for g, _ in self._outs if g is not None}
        in_place = ringed and len(gains) == 1 and all(g is not None for g, _ in self._outs)
        writes = []
        scaled = {}
        for gain, sink in self._outs:
            out = iq
            if gain is not None:
                k = complex(gain.k())
                if k not in scaled:
                    out = iq if in_place else torch.empty_like(iq)
                    raw._lib.check(ctx._lib.ofdmtx_scale_cc(ctx._h, raw._ptr(iq), iq.numel(), k.real, k.imag, raw._ptr(out),
                                                            raw._stream(ctx.device)))
                    scaled[k] = out
                out = scaled[k]
            writes.append((sink, out))
        self._output.hand_over(writes, ringed)

    @staticmethod
    def add_options(normal, expert):
        normal.add_option("-s", "--size", type="int", default=16, help="OFDM symbols per frame [default=%default]")
        raw.ofdm_params.add_options(normal, expert)


def make_data(data_tones, num_symbols, filename=None):
    """ofdm_rxtx.py:101-121: num_symbols * data_tones values from a complex64 file, or unit QPSK
    points exp(j pi/4 (2k+1)), k uniform in 0..3, from the generator seeded with 78532; returned as
    a repeating vector source of data_tones values per item.

    The reference draws k with Python 2's random.randint, which is int(random() * 4) there; Python 3's
    randint consumes the generator differently, so the Python 2 rule is spelled out to keep the data."""
    count = int(num_symbols) * int(data_tones)
    if filename is not None:
        data = np.fromfile(filename, dtype=np.complex64, count=count)
    else:
        stream = random.Random(78532)
        nn = 4
        data = np.array([cmath.exp(1j * (math.pi / nn * (2 * int(stream.random() * nn) + 1))) for _ in range(count)],
                        dtype=np.complex64)
    return gr.vector_source_c(data, repeat=True, vlen=data_tones)


class sender_thread(object):
    """ofdm_rxtx.py:179-215: sends `numpkts` frames in bursts of `burst`, then end of stream, and
    reports symbols per second."""

    def __init__(self, tx, options):
        self.tx = tx
        self.pause = options.pause
        self.numpkts = options.numpkts
        self.burst = options.burst

    @staticmethod
    def add_options(parser):
        parser.add_option("-N", "--numpkts", type="float", default=0, help="frames to transmit [default=%default]")
        parser.add_option("", "--pause", type="float", default=0, help="pause between bursts, seconds")
        parser.add_option("", "--burst", type="int", default=1, help="frames per burst")

    def run(self):
        start = time.time()
        numpkts = abs(int(self.numpkts))
        if numpkts == 0:
            raise ValueError("sender_thread: a batch modulator needs a finite --numpkts")   # the reference sends forever
        pktno = 0
        while pktno < numpkts:                      # (the reference tests !=, overshooting when burst does not divide)
            self.tx.send_packet(self.burst)
            pktno += self.burst
            if self.pause > 0.0:
                time.sleep(self.pause)
        self.tx.send_eof()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        elapsed = time.time() - start
        sys.stdout.write("Sent %d packets of %d symbols in %.2f seconds\n%s\n"
                         % (pktno, self.tx.size, elapsed, pktno * self.tx.size / max(elapsed, 1e-9)))
        return pktno
