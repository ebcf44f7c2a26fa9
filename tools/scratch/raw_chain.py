"""rawofdm bytes -> carrier values -> IQ: one pass over the whole batch against cache-sized chunks
(the carrier values of a chunk then never leave the Infinity Cache)."""
import numpy as np, torch
from ofdm_amd.batch import TxContext
from ofdm_amd import raw

ctx = TxContext(0)
dev = ctx.device

class O(object):
    num_preambles, half_sync, verbose, log = 2, True, False, False

def run(bitrate, nfft, cp, n_frames, nsym=16):
    o = O(); o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), cp
    p = raw.ofdm_params(o)
    tx = raw.qam_tx(bitrate, p.data_tones, nsym, ctx=ctx)
    data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=dev)
    row = p.fft_length + p.cp_length
    iq = torch.empty(n_frames * (nsym + 3) * row, dtype=torch.complex64, device=dev)
    res = {}
    for chunk in (n_frames, 8192, 4096, 2048, 1024, 512):
        if chunk > n_frames: continue
        plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, chunk, nsym)
        vals = torch.empty(chunk * nsym * p.data_tones, dtype=torch.complex64, device=dev)
        def once():
            for c0 in range(0, n_frames, chunk):
                tx.modulate(data[c0 * tx.framebytes:(c0 + chunk) * tx.framebytes], out=vals)
                plan.modulate(vals, out=iq[c0 * (nsym + 3) * row:(c0 + chunk) * (nsym + 3) * row])
        for _ in range(2): once()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): once()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        res[chunk] = ms
        inter_mb = chunk * nsym * p.data_tones * 8 / 1e6
        print("bitrate %d N=%d cp=%d: chunk %6d frames (%6.1f MB of carrier values): %.3f ms, %.0f Gsamples/s"
              % (bitrate, nfft, cp, chunk, inter_mb, ms, iq.numel() / ms / 1e6), flush=True)
    return res

run(8, 256, 64, 65536)
run(5, 256, 64, 65536)
run(8, 1024, 256, 16384)
