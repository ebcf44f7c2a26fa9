import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ofdm_amd import raw
from ofdm_amd.batch import TxContext
ctx = TxContext(0); dev = ctx.device
class O: num_preambles, half_sync, verbose, log = 2, True, False, False
for nfft, cp in ((256, 64), (1024, 256), (64, 16)):
    o = O(); o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125*nfft), cp
    p = raw.ofdm_params(o)
    n_sym = (1 << 22) if nfft <= 256 else (1 << 20)
    plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_sym // 16, 16)
    data = torch.randn(n_sym * p.data_tones, dtype=torch.complex64, device=dev)
    iq = torch.empty(plan.n_samples, dtype=torch.complex64, device=dev)
    for _ in range(2): plan.modulate(data, out=iq)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); plan.modulate(data, out=iq); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    by = data.numel()*8 + plan.n_samples*8
    print("N=%d cp=%d dbg=%s: %.3f ms  %.0f GB/s (%.1f%%)" % (nfft, cp, os.environ.get("RAW_DBG","0"), ms, by/ms/1e6, by/ms/1e6/80))
