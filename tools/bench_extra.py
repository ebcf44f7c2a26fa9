#!/usr/bin/env python3
"""Secondary measurements (not the headline bench line): the mixed-rate ragged batch and the
variable-size modulator, each as one JSON line with the kernel time from events on the launch
stream and the algorithmic HBM bytes per launch.

    python tools/bench_extra.py [--steps K]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ofdm_amd import raw  # noqa: E402
from ofdm_amd.batch import TxContext  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


# Output buffers: one arena of device memory whose 64 MiB pieces alternate between the three parts of the device
# memory (TxContext.alloc_spread; --placement first: ordinary allocations), carved up per workload.
ARENA = {"mem": None, "mode": "first"}


def out_buffer(ctx, n_samples):
    need = int(n_samples) * 8
    if ARENA["mem"] is not None and ARENA["mem"].numel() >= need:
        return ARENA["mem"][:need].view(torch.complex64)
    return torch.empty(int(n_samples), dtype=torch.complex64, device=ctx.device)


def line(name, ms, alg_bytes, samples, extra):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    out = {"workload": name, "kernel_ms": ms, "Msamples_per_s": samples / (ms * 1e-3) / 1e6,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes}}
    out["output_placement"] = ARENA["mode"]
    out.update(extra)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--ragged-packets", type=int, default=400_000)
    ap.add_argument("--raw-symbols", type=int, default=1 << 20)
    ap.add_argument("--only", choices=["qam", "chain"], default=None, help="run one section only")
    ap.add_argument("--placement", choices=["spread", "first"], default="spread")
    ap.add_argument("--arena-gb", type=float, default=90.0, help="size of the spread output arena")
    a = ap.parse_args()
    ctx = TxContext(0)
    dev = ctx.device
    if a.placement == "spread":
        mem, spread = ctx.alloc_spread(int(a.arena_gb * 1e9) // 8 * 8, parts=3)
        if spread:
            ARENA["mem"], ARENA["mode"] = mem, "spread"
    if a.only == "qam":
        return bench_qam_tx(ctx, a.steps)
    if a.only == "chain":
        return bench_raw_chain(ctx, a.steps)

    # mixed-rate ragged batch: all eight rates, payload lengths 1..2304 bytes
    rng = np.random.default_rng(3)
    n = a.ragged_packets
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(1, 2305, n).astype(np.uint32)
    payloads = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=dev)
    plan = ctx.plan_ragged(regimes, lens)
    iq = out_buffer(ctx, plan.n_samples)
    ms = timed(lambda: plan.tx(payloads, out=iq), a.steps)
    line("mixed-rate ragged batch, payload -> IQ (fused kernel)", ms, int(lens.sum()) + plan.n_samples * 8,
         plan.n_samples, {"packets": n})
    msg = plan.alloc_msg()
    ms_e = timed(lambda: plan.encode_bits(payloads, out=msg), a.steps)
    ms_m = timed(lambda: plan.modulate(msg, out=iq), a.steps)
    ms_rx = timed(lambda: plan.loopback(iq), max(1, a.steps // 2))
    print(json.dumps({"workload": "loop-back verifier on the mixed-rate ragged batch (demod + decode)", "total_ms": ms_rx,
                      "Msamples_per_s": plan.n_samples / (ms_rx * 1e-3) / 1e6, "packets": n}), flush=True)
    line("mixed-rate ragged batch, mapper messages -> IQ (k_tx, message input)", ms_m, plan.msg_bytes + plan.n_samples * 8,
         plan.n_samples, {"packets": n, "k_encode_bits_ms": ms_e})
    del iq, msg, payloads, plan

    # short packets (ACK-sized .. 100 bytes), all rates: many packets per output byte
    n = 2_000_000
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 101, n).astype(np.uint32)
    payloads = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=dev)
    plan = ctx.plan_ragged(regimes, lens)
    iq = out_buffer(ctx, plan.n_samples)
    ms = timed(lambda: plan.tx(payloads, out=iq), a.steps)
    msg = plan.alloc_msg()
    ms_e = timed(lambda: plan.encode_bits(payloads, out=msg), a.steps)
    line("short packets (0..100 bytes), mixed rates, payload -> IQ (fused kernel)", ms, int(lens.sum()) + plan.n_samples * 8,
         plan.n_samples, {"packets": n, "k_encode_bits_ms": ms_e})
    del iq, msg, payloads, plan

    # uniform batches of the other rates, fused kernel
    for regime in (1, 3, 5, 7):
        npk = 200_000 if regime == 1 else 500_000
        plan = ctx.plan_uniform(npk, regime, 1500)
        payloads = torch.randint(0, 256, (npk * 1500,), dtype=torch.uint8, device=dev)
        iq = out_buffer(ctx, plan.n_samples)
        ms = timed(lambda: plan.tx(payloads, out=iq), a.steps)
        line("regime %d, %d x 1500-byte packets, payload -> IQ (fused kernel)" % (regime, npk), ms,
             npk * 1500 + plan.n_samples * 8, plan.n_samples, {"packets": npk})
        del iq, payloads, plan

    # variable-size modulator: frames of 16 data symbols + 3 preamble symbols, unit QPSK
    class O(object):
        num_preambles, half_sync, verbose, log = 2, True, False, False
    for nfft, cp in ((256, 64), (256, 32), (256, 16), (1024, 256), (1024, 128), (1024, 16), (64, 16)):
        o = O()
        o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), cp
        p = raw.ofdm_params(o)
        n_sym = a.raw_symbols if nfft <= 256 else a.raw_symbols // 4
        n_frames = n_sym // 16
        plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, 16)
        ph = torch.randint(0, 4, (n_frames * 16 * p.data_tones,), device=dev)
        data = torch.polar(torch.ones_like(ph, dtype=torch.float32), (2 * ph + 1) * (np.pi / 4)).contiguous()
        iq = out_buffer(ctx, plan.n_samples)
        ms = timed(lambda: plan.modulate(data, out=iq), a.steps)
        line("variable-size modulator N=%d cp=%d, %d data symbols in frames of 16 (+3 preamble rows)"
             % (nfft, cp, n_frames * 16), ms, data.numel() * 8 + plan.n_samples * 8, plan.n_samples,
             {"data_symbols": n_frames * 16})
        del iq, data, plan
    bench_qam_tx(ctx, a.steps)
    bench_raw_chain(ctx, a.steps)


def bench_qam_tx(ctx, steps):
    """bytes -> carrier values (the bit chain in front of the variable-size modulator)"""
    dev = ctx.device
    for bitrate, ncar in ((8, 192), (8, 768), (5, 192), (1, 192)):
        tx = raw.qam_tx(bitrate, ncar, 16, ctx=ctx)
        n_frames = (1 << 20) // 16 if ncar == 192 else (1 << 18) // 16
        data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=dev)
        out = out_buffer(ctx, n_frames * 16 * ncar)
        ms = timed(lambda: tx.modulate(data, out=out), steps)
        line("rawofdm qam_tx bitrate %d, %d carriers x 16 symbols per frame, %d frames" % (bitrate, ncar, n_frames),
             ms, data.numel() + out.numel() * 8, out.numel(), {"frames": n_frames})


def bench_raw_chain(ctx, steps):
    """bytes -> IQ for rawofdm: the two kernels with the carrier values in memory against the one-kernel path"""
    dev = ctx.device

    class O(object):
        num_preambles, half_sync, verbose, log = 2, True, False, False
    for bitrate, nfft, cp in ((8, 256, 64), (5, 256, 64), (1, 256, 64), (8, 128, 32), (8, 512, 128), (8, 1024, 256), (8, 64, 16), (3, 64, 16)):
        o = O()
        o.fft_length, o.occupied_tones, o.cp_length = nfft, int(0.8125 * nfft), cp
        mod = raw.ofdm_mod(o, ctx=ctx)
        p = mod.params
        nsym = 16
        n_frames = ((1 << 20) if nfft <= 256 else (1 << 18)) // nsym
        tx = raw.qam_tx(bitrate, p.data_tones, nsym, ctx=ctx)
        data = torch.randint(0, 256, (n_frames * tx.framebytes,), dtype=torch.uint8, device=dev)
        plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, nsym)
        iq = out_buffer(ctx, plan.n_samples)
        vals = torch.empty(n_frames * nsym * p.data_tones, dtype=torch.complex64, device=dev)

        def two():
            tx.modulate(data, out=vals)
            plan.modulate(vals, out=iq)
        ms2 = timed(two, steps)
        ms1 = timed(lambda: mod.modulate_bytes(bitrate, data, nsym, out=iq), steps)
        line("rawofdm bytes -> IQ in one kernel, bitrate %d, N=%d cp=%d, %d frames of %d symbols" % (bitrate, nfft, cp, n_frames, nsym),
             ms1, data.numel() + plan.n_samples * 8, plan.n_samples, {"two_kernels_ms": ms2, "fused": bool(mod.fused)})


if __name__ == "__main__":
    main()
