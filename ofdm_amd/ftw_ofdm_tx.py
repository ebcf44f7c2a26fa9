#!/usr/bin/env python3
"""Counterpart of the reference's ftw_80211_ofdm_tx/src/examples/ftw_ofdm_tx_uhd.py without the
radio: the same options and control flow, the transmit path on the GPU, output to --to-file (raw
little-endian float32 I,Q pairs, examples/ftw_ofdm_tx_uhd.py:66-67) or to a null sink.

    python -m ofdm_amd.ftw_ofdm_tx --to-file out.dat --regime 8 -W 20e6 -N 1000 -r 150
"""
from optparse import OptionParser

from . import gr
from .ftw_ofdm import ftw_transmit_path


class my_top_block(gr.top_block):
    """examples/ftw_ofdm_tx_uhd.py:56-89."""

    def __init__(self, options, payload=''):
        gr.top_block.__init__(self)
        if getattr(options, "tx_freq", None) is not None:
            raise NotImplementedError("radio sinks are outside this library's scope; use --to-file")
        elif options.to_file is not None:
            self.sink = gr.file_sink(gr.sizeof_gr_complex, options.to_file)
        else:
            self.sink = gr.null_sink(gr.sizeof_gr_complex)
        options.interp = 100e6 / options.bandwidth          # FTW-specific conversion (ftw_ofdm_tx_uhd.py:71)
        if getattr(options, "from_file", None) is None:
            self.txpath = ftw_transmit_path(options, payload)
        else:
            self.txpath = gr.file_source(gr.sizeof_gr_complex, options.from_file)
        self._tx_amplitude = options.tx_amplitude
        self.amp = gr.multiply_const_cc(self._tx_amplitude)
        self.connect(self.txpath, self.amp, self.sink)
        if getattr(options, "log", False):                  # ftw_ofdm_tx_uhd.py:88-89: the path's output in front of tx_amplitude
            self.connect(self.txpath, gr.file_sink(gr.sizeof_gr_complex, 'ftw_benchmark.dat'))


def make_parser():
    parser = OptionParser(conflict_handler="resolve")
    parser.add_option("", "--from-file", default=None, help="use intput file for packet contents")
    parser.add_option("", "--to-file", default=None, help="Output file for modulated samples")
    parser.add_option("-s", "--swapIQ", action="store_true", default=False,
                      help="swap IQ components before sending to the sink [default=%default]")
    parser.add_option("-v", "--verbose", action="store_true", default=False)
    parser.add_option("-l", "--log", action="store_true", default=False,
                      help="write debug-output of individual stages to disk")
    parser.add_option("-W", "--bandwidth", type="float", default=500e3,
                      help="set symbol bandwidth [default=%default] 20e6 -> 802.11a/g, OFDM-symbolduration=4us, "
                           "10e6 -> 802.11p, OFDM-symbolduration=8us")
    parser.add_option("", "--regime", type="string", default="1",
                      help="set OFDM coderegime 1..8: 6, 9, 12, 18, 24, 36, 48, 54 Mbit/s at 20 MHz [default=%default]")
    parser.add_option("", "--tx-amplitude", type="float", default=0.3,
                      help="set gain factor for complex baseband floats [default=%default]")
    parser.add_option("-N", "--num", type="int", default=1, help="set number of packets to send, [default=%default]")
    parser.add_option("-r", "--repeat", type="int", default=1,
                      help="set number of HelloWorld in single packet to send, [default=%default]")
    parser.add_option("-p", "--payload", type="string", default=None,
                      help="payload ASCII-string to send, [default=%default]")
    parser.add_option("-R", "--role", type="string", default=None, help="PNC role: A or B [default=%default]")
    return parser


def main(argv=None):
    (options, args) = make_parser().parse_args(argv)
    if options.role == 'B':                                 # ftw_ofdm_tx_uhd.py:143-148
        options.payload = "WorldHello" * options.repeat
    else:
        options.payload = "HelloWorld" * options.repeat
    my_msg = options.payload
    tb = my_top_block(options, my_msg)
    gr.enable_realtime_scheduling()
    tb.start()
    if options.from_file is None:
        counter = 0
        while counter < options.num:
            tb.txpath.send_pkt(my_msg, eof=False)
            counter = counter + 1
        print("End of Transmission")
        tb.txpath.send_pkt(eof=True)
    tb.wait()
    return tb


if __name__ == '__main__':
    try:
        main()
    except KeyboardInterrupt:
        pass
