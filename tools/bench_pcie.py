"""Host buffers in, host buffers out: the rate a caller sees who hands over payloads in host memory
and wants the samples back in host memory (the `--to-file` flow of the drop-in scripts), as opposed
to bench.py's device-resident figure.

Chunks of 64 Ki packets: payloads host -> device, ofdmtx_tx, samples device -> host, through two
page-locked buffers per direction and a copy stream, so that the copy back of chunk i overlaps the
kernel of chunk i+1.  48 000 bytes leave the GPU for every 1 500 that enter: the link rate of the
copy back bounds the result.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofdm_amd.batch import TxContext  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=16)
    ap.add_argument("--chunk-packets", type=int, default=65536)
    ap.add_argument("--regime", type=int, default=8)
    ap.add_argument("--payload", type=int, default=1500)
    a = ap.parse_args()

    ctx = TxContext(0)
    dev = ctx.device
    plan = ctx.plan_uniform(a.chunk_packets, a.regime, a.payload)
    n_in = a.chunk_packets * a.payload
    host_in = [torch.randint(0, 256, (n_in,), dtype=torch.uint8).pin_memory() for _ in range(2)]
    host_out = [torch.empty(plan.n_samples, dtype=torch.complex64).pin_memory() for _ in range(2)]
    d_in = [torch.empty(n_in, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_iq = [plan.alloc_iq() for _ in range(2)]
    main_s = torch.cuda.current_stream(dev)
    copy_s = torch.cuda.Stream(device=dev)

    def run(n_chunks):
        done = [None, None]      # copy back of the buffer finished
        for i in range(n_chunks):
            b = i & 1
            if done[b] is not None:
                main_s.wait_event(done[b])                 # d_iq[b] and host_out[b] free again
            d_in[b].copy_(host_in[b], non_blocking=True)
            plan.tx(d_in[b], out=d_iq[b])
            ready = torch.cuda.Event()
            ready.record(main_s)
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(ready)
                host_out[b].copy_(d_iq[b], non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record(copy_s)
        torch.cuda.synchronize(dev)

    run(2)
    t0 = time.perf_counter()
    run(a.chunks)
    dt = time.perf_counter() - t0
    samples = a.chunks * plan.n_samples
    # device-resident kernel time for the same chunk, for the ratio
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(4):
        plan.tx(d_in[0], out=d_iq[0])
    ev1.record()
    torch.cuda.synchronize(dev)
    k_ms = ev0.elapsed_time(ev1) / 4
    print(json.dumps({
        "workload": "regime %d, %d B payloads, %d chunks of %d packets, host to host" % (a.regime, a.payload, a.chunks, a.chunk_packets),
        "samples_per_s_host_to_host": samples / dt,
        "bytes_back_per_s": samples * 8 / dt,
        "seconds": dt,
        "kernel_ms_per_chunk": k_ms,
        "samples_per_s_device_resident": plan.n_samples / (k_ms * 1e-3)}))


if __name__ == "__main__":
    main()
