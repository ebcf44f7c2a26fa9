"""End-to-end rate of the drop-in path: host payload bytes in (send_pkt), IQ to a sink.  The first batch of a
transmit path pays for its set-up (the output ring is placed then: ofdmtx_stream_alloc looks for memory of
different parts of the device); it is timed on its own, the rate is that of the batches behind it."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ofdm_amd import batch, ftw_ofdm_tx, gr

class O: pass
o = O(); o.regime="8"; o.bandwidth=20e6; o.swapIQ=False; o.log=False; o.role=None; o.tx_amplitude=0.3
o.to_file=None; o.from_file=None; o.tx_freq=None
rng = np.random.default_rng(0)
n = 65536 * 2
pay = [rng.integers(0,256,1500,dtype=np.uint8).tobytes() for _ in range(64)]
for sink_kind in ("null", "pageable-host", "file:/dev/null", "file:/dev/null", "file:/dev/null without the ring", "null without the ring"):
    batch.StreamOutput.RING_MIN_BYTES = (1 << 60) if "without" in sink_kind else (256 << 20)
    o.to_file = "/dev/null" if sink_kind.startswith("file") else None
    tb = ftw_ofdm_tx.my_top_block(o, pay[0])
    if sink_kind == "pageable-host":
        class HostSink:
            def __init__(s): s.n = 0
            def write(s, t): s.n += t.cpu().numel()
            def close(s): pass
        tb.connect(tb.txpath, tb.amp, HostSink())
    tb.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(65536): tb.txpath.send_pkt(pay[i & 63])          # the first batch: set-up included
    tb.txpath.ofdm_tx._output.drain()                                # (a batch reaches its sinks one batch late: hand it over now)
    torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    ring = tb.txpath.ofdm_tx._output.ring
    t0 = time.perf_counter()
    for i in range(n): tb.txpath.send_pkt(pay[i & 63])
    tb.txpath.send_pkt(eof=True); tb.wait()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s sink: first batch %.3f s (ring: %s); then %d packets in %.3f s = %.1f Msamples/s (%.2f GB/s of IQ)"
          % (sink_kind, t_first, "none" if ring is None else "%d parts, %.0f GiB held while placed" % (ring.parts, ring.peak_held_bytes / 2**30),
             n, dt, n*6000/dt/1e6, n*48000/dt/1e9))
    del tb, ring
