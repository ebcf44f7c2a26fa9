#!/usr/bin/env python3
"""Headline benchmark: 802.11a 54 Mbps, 1500-byte packets, payload -> full PPDU IQ.

    python bench.py --gpus N --steps K --warmup W [--packets P]
    python bench.py --workload rawofdm [--raw-n 1024 --raw-cp 256 --raw-symbols 262144]   (secondary)
    python bench.py --workload ragged [--ragged-packets 400000]                            (secondary)

One "step" is one pass of the transmit chain over one batch of P synthetic packets per GPU,
payloads already resident in HBM: by default the fused kernel k_tx (payload -> IQ, the mapper
bits never leave the CU); --fused 0 times the two-kernel path (k_encode_bits writes the mapper
messages, the message-input instantiation of k_tx turns them into IQ).  With N > 1 there is one
rank (process) per GPU and the batch is sharded by packet index: every rank processes its own P
packets, there is no collective on the data path, and the reported value is all ranks' samples
divided by the slowest rank's time (weak scaling).

How the N ranks come to be: under a launcher (torch.distributed.run sets WORLD_SIZE / RANK /
LOCAL_RANK / MASTER_*) this process IS one rank.  Started plainly as `python bench.py --gpus N`
with N > 1 it starts the N ranks itself: the parent parses the arguments, imports nothing that
could touch a GPU, starts N fresh children of this same file with the rank environment set
(rendezvous on 127.0.0.1), relays rank 0's JSON line and exits with the first non-zero child
status.  `--gpus 1` runs in this process.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_tx; with --fused 0 its
message-input instantiation) against the HBM peak from its algorithmic bytes and its launch time measured with
events on the launch stream; `cpu_baseline` times the plain-C oracle port on the host cores (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy, torch and the library binding are imported by load_runtime(), i.e. only in a process that is a
# rank: the parent of a self-started multi-rank run must stay free of anything that initialises a GPU
np = torch = _lib = TxContext = None


def load_runtime():
    global np, torch, _lib, TxContext
    import numpy
    import torch as torch_
    from ofdm_amd import _lib as lib_
    from ofdm_amd.batch import TxContext as ctx_
    np, torch, _lib, TxContext = numpy, torch_, lib_, ctx_


REGIME = 8            # 54 Mbit/s, 64-QAM r=3/4
PAYLOAD = 1500
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def cpu_baseline(n_sample, threads):
    """Plain-C oracle port of the same chain on the host (payload -> IQ), packet-parallel."""
    from oracle import c_oracle
    info = _lib.packet_info(REGIME, PAYLOAD)
    rng = np.random.default_rng(2)
    payloads = rng.integers(0, 256, n_sample * PAYLOAD, dtype=np.uint8)
    idx = np.arange(n_sample, dtype=np.int64)
    samples = info.n_rows * 80
    args = dict(pay_off=idx * PAYLOAD, pay_len=np.full(n_sample, PAYLOAD, np.int32),
                regimes=np.full(n_sample, REGIME, np.uint8),
                pilot_off=((idx * (info.n_sym + 1)) % 127).astype(np.int32),
                msg_off=idx * info.msg_len, iq_off=idx * samples,
                msg_total=n_sample * info.msg_len, iq_total=n_sample * samples,
                amp=0.3, flags=3, threads=threads, want_msg=False, want_iq=True)
    c_oracle.tx_batch(payloads[:PAYLOAD * 64], args["pay_off"][:64], args["pay_len"][:64], args["regimes"][:64],
                      args["pilot_off"][:64], args["msg_off"][:64], args["iq_off"][:64],
                      64 * info.msg_len, 64 * samples, threads=1, want_msg=False)   # warm caches / page in
    t0 = time.perf_counter()
    c_oracle.tx_batch(payloads, **args)
    dt = time.perf_counter() - t0
    # one thread on a proportionally smaller sample, for the per-core figure
    n1 = max(64, n_sample // threads)
    a1 = dict(args, pay_off=args["pay_off"][:n1], pay_len=args["pay_len"][:n1], regimes=args["regimes"][:n1],
              pilot_off=args["pilot_off"][:n1], msg_off=args["msg_off"][:n1], iq_off=args["iq_off"][:n1],
              msg_total=n1 * info.msg_len, iq_total=n1 * samples, threads=1)
    t1 = time.perf_counter()
    c_oracle.tx_batch(payloads[:n1 * PAYLOAD], **a1)
    dt1 = time.perf_counter() - t1
    return {"value": n_sample * samples / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "value_one_core": n1 * samples / dt1 / 1e6,
            "sample": "%d packets (54 Mbps, 1500 B) payload->IQ by oracle/ftw_oracle.c, %.1f s wall on %d threads; "
                      "restatement, not GNU Radio" % (n_sample, dt, threads)}


def py_refshaped_baseline():
    """Config 1 (one 1500-byte packet at 6 Mbit/s) through the per-bit pure-Python restatement that keeps
    the control flow of the reference's send_pkt (oracle/ftw_pybits.py): the order of magnitude of
    the reference's Python-bound bit chain, one core."""
    from oracle import ftw_pybits
    payload = np.random.default_rng(1).integers(0, 256, PAYLOAD, dtype=np.uint8).tobytes()
    t0 = time.perf_counter()
    ftw_pybits.encode_bits(payload, "1", 4.0)
    dt = time.perf_counter() - t0
    samples = _lib.packet_info(1, PAYLOAD).n_rows * 80
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "config 1: one 1500-byte packet at 6 Mbit/s (42 320 samples), bit chain only (no modulation), "
                      "per-bit Python in the shape of the reference's send_pkt, %.1f ms; restatement, not the "
                      "reference" % (dt * 1e3)}


def host_threads():
    """Threads for the CPU baseline: every core this job may use.  The scheduler's affinity mask and
    the thread limit the box exports (OMP_NUM_THREADS: the job's CPU share) bound it; os.cpu_count()
    is the whole machine, which a one-GPU job does not own."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def load_traffic(workload_key):
    """HBM bytes per launch of the named kernel from the committed PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get(workload_key)
    except (OSError, ValueError):
        return None


def alloc_output(ctx, n_samples, mode, fill, candidates=5, compare=True, inputs=None, move_inputs=True):
    """The output buffer of a workload and a record of how it was placed.  Where a large buffer lies in
    HBM changes the store rate on MI355X (one of the memory's three parts: 5.6-5.7 TB/s, pieces
    alternating between the parts: 7.1 TB/s), so by default the buffer comes from
    TxContext.alloc_spread; one launch into an ordinary first allocation is timed beside it for the
    record.  mode "probe": the fastest of a few ordinary allocations; "first": the first allocation
    as it comes.  fill(buffer) is the workload's own launch.
    inputs: the workload's input tensor.  With a spread buffer it is moved (unless move_inputs is false) into
    the tail of the same buffer -- pieces of all parts, like the output: the headline launch reads 1.5 GB
    beside 48 GB of stores and runs 0.9 % faster with them there (tools/scratch/payload_in_spread.py), the
    variable-size modulator, a third of whose traffic is reads, 2 %.  The third value returned is the tensor
    to use from then on (the one passed in where nothing was moved)."""
    dev = ctx.device
    grain = 64 << 20
    out_bytes = -(-n_samples * 8 // grain) * grain
    in_bytes = inputs.numel() * inputs.element_size() if (inputs is not None and move_inputs) else 0

    def once(b):
        fill(b)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        fill(b)
        t1.record()
        t1.synchronize()
        return float(t0.elapsed_time(t1))

    if mode == "spread" and os.environ.get("BENCH_BACKEND", "nccl") != "nccl":
        mode = "first"                                    # rehearsal with ranks sharing a device: no exclusive use of its memory
    if mode == "spread":
        # (a device whose memory a process before has just given back can look full, or hand out memory in
        #  no order at all, for a few seconds: a second and a third attempt, two seconds apart)
        attempts = 0
        while True:
            attempts += 1
            t_alloc = time.perf_counter()
            try:
                raw, spread = ctx.alloc_spread(out_bytes + in_bytes, parts=3)
            except _lib.OfdmTxError:                      # (e.g. another process took the memory meanwhile)
                raw, spread = torch.empty(n_samples * 8, dtype=torch.uint8, device=dev), False
            t_alloc = time.perf_counter() - t_alloc
            if spread or attempts == 3:
                break
            del raw
            torch.cuda.empty_cache()
            time.sleep(2.0)
        iq = raw[:n_samples * 8].view(torch.complex64)
        if spread:
            if in_bytes:
                moved = raw[out_bytes:out_bytes + in_bytes].view(inputs.dtype).view(inputs.shape)
                moved.copy_(inputs)
                inputs = moved
            rec = {"mode": "spread", "note": "64 MiB pieces alternating between `parts` of the device memory's three parts, "
                                             "found by timing writes into the memory taken (ofdmtx_stream_alloc; peak_held_GB "
                                             "= what it held while it looked); one "
                                             "launch timed into it and (unless --placement-compare 0) into an ordinary first "
                                             "allocation, which is freed before the timed region",
                   "alloc_s": round(t_alloc, 3), "attempts": attempts, "peak_held_GB": round(ctx.last_alloc["peak_held_bytes"] / 1e9, 1),
                   "handles": ctx.last_alloc["handles"], "parts": ctx.last_alloc["parts"], "spread_ms": round(once(iq), 3)}
            rec["kept"] = "spread"
            rec["inputs"] = "in the tail of the same buffer (pieces of all parts)" if in_bytes else "an ordinary allocation"
            if compare:
                plain = torch.empty(n_samples, dtype=torch.complex64, device=dev)
                rec["first_allocation_ms"] = round(once(plain), 3)
                if rec["first_allocation_ms"] < rec["spread_ms"]:
                    iq, rec["kept"] = plain, "first allocation"
                del plain
                torch.cuda.empty_cache()
            return iq, rec, inputs
        del iq, raw
        mode = "probe"                                    # the device could not provide the layout
    if mode == "probe":
        cands, times = [], []
        need = n_samples * 8
        while len(cands) < max(1, candidates):
            free, _ = torch.cuda.mem_get_info(dev)
            if cands and free < need + (8 << 30):
                break
            cands.append(torch.empty(n_samples, dtype=torch.complex64, device=dev))
        times = [once(b) for b in cands]
        keep = int(np.argmin(times))
        iq = cands[keep]
        del cands
        torch.cuda.empty_cache()
        return iq, {"mode": "probe", "candidates_ms": [round(t, 3) for t in times], "chosen": keep,
                    "note": "one launch timed into each candidate buffer (allocation order); the fastest is kept, "
                            "the others freed before the timed region"}, inputs
    return torch.empty(n_samples, dtype=torch.complex64, device=dev), {"mode": "first"}, inputs


def timed_steps(step, a, dist, backend, dev):
    """W untimed steps, then K timed ones between barriers; returns (seconds, max over ranks; events)."""
    for _ in range(a.warmup):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(evs[i])          # events ride on the stream the kernels are launched on
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, evs


def run_rawofdm(a, ctx, dev, world, rank, dist, backend):
    """Config 4: rawofdm's variable-size modulator (carrier values -> IQ), frames sharded by index."""
    from ofdm_amd import raw

    class O(object):
        num_preambles, half_sync, verbose, log = 2, True, False, False
    o = O()
    o.fft_length, o.occupied_tones, o.cp_length = a.raw_n, int(0.8125 * a.raw_n), a.raw_cp
    p = raw.ofdm_params(o)
    size = 16
    n_frames = a.raw_symbols // size
    plan = raw.plan_frames(ctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, size)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4 + rank)
    ph = torch.randint(0, 4, (n_frames * size * p.data_tones,), device=dev, generator=gen)
    data = torch.polar(torch.ones_like(ph, dtype=torch.float32), (2 * ph + 1) * (np.pi / 4)).contiguous()   # unit QPSK
    iq, placement, data = alloc_output(ctx, plan.n_samples, a.placement, lambda b: plan.modulate(data, out=b), a.placement_candidates,
                                       bool(a.placement_compare), inputs=data, move_inputs=bool(a.inputs_spread))

    def step(events=None):
        if events is not None:
            events[0].record()
        plan.modulate(data, out=iq)
        if events is not None:
            events[1].record()
            events[2].record()

    dt, evs = timed_steps(step, a, dist, backend, dev)
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    by_rank = per_rank(kernel_ms, dist, backend, dev, world)
    if rank == 0:
        alg_bytes = data.numel() * 8 + plan.n_samples * 8          # carrier values read + rows written
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "baseband IQ Msamples/s (node), rawofdm N=%d cp=%d" % (a.raw_n, a.raw_cp),
            "value": world * plan.n_samples * a.steps / dt / 1e6, "unit": "Msamples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "rawofdm modulator N=%d cp=%d, %d occupied tones (%d data), %d frames of %d data symbols "
                                   "+ %d preamble symbols per GPU, unit QPSK values -> IQ"
                                   % (a.raw_n, a.raw_cp, o.occupied_tones, p.data_tones, n_frames, size, plan.n_pre),
                       "parallelism": "frames sharded by index over %d GPU(s), no collective" % world},
            "iq_placement": placement,
            "kernel_ms": {"k_raw_modulate": kernel_ms},
            "ranks": ranks_record(a, dist, backend, world, by_rank),
            "roofline": {"bound": "hbm", "kernel": "k_raw_modulate", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes}}), flush=True)


def run_ragged(a, ctx, dev, world, rank, dist, backend):
    """Config 3: mixed-rate ragged batch (all eight rates, payloads of 1..2304 bytes), packet ranges per rank."""
    n = a.ragged_packets
    n_dbps = np.array([0, 24, 36, 48, 72, 96, 144, 192, 216])

    def shard(r):
        rng = np.random.default_rng([3, r])
        return rng.integers(1, 9, n).astype(np.uint8), rng.integers(1, 2305, n).astype(np.uint32)

    # the stream-wide pilot counter runs on over the shards in front of this one (SIGNAL + DATA symbols each)
    symbols_before = 0
    for r in range(rank):
        rg, ln = shard(r)
        bits = 16 + 8 * (ln.astype(np.int64) + 28) + 6
        symbols_before += int(((bits + n_dbps[rg] - 1) // n_dbps[rg] + 1).sum())
    regimes, lens = shard(rank)
    plan = ctx.plan_ragged(regimes, lens, pilot_start=symbols_before % 127)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3 + rank)
    payloads = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=dev, generator=gen)
    iq, placement, payloads = alloc_output(ctx, plan.n_samples, a.placement, lambda b: plan.tx(payloads, out=b), a.placement_candidates,
                                           bool(a.placement_compare), inputs=payloads, move_inputs=bool(a.inputs_spread))

    def step(events=None):
        if events is not None:
            events[0].record()
        plan.tx(payloads, out=iq)
        if events is not None:
            events[1].record()
            events[2].record()

    dt, evs = timed_steps(step, a, dist, backend, dev)
    samples = torch.tensor([plan.n_samples], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist:
        dist.all_reduce(samples)                       # shards differ in size: the job's samples per step
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    by_rank = per_rank(kernel_ms, dist, backend, dev, world)
    if rank == 0:
        alg_bytes = int(lens.sum()) + plan.n_samples * 8            # payloads read + rows written (rank 0's shard)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "baseband IQ Msamples/s (node), 802.11a mixed-rate ragged batch",
            "value": float(samples.item()) * a.steps / dt / 1e6, "unit": "Msamples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "802.11a, all eight rates, payloads of 1..2304 bytes, %d packets per GPU, payload -> "
                                   "full PPDU IQ" % n,
                       "packets_per_gpu": n,
                       "parallelism": "packet ranges per rank over %d GPU(s), pilot counter continued, no collective" % world},
            "iq_placement": placement,
            "kernel_ms": {"k_tx": kernel_ms},
            "ranks": ranks_record(a, dist, backend, world, by_rank),
            "roofline": {"bound": "hbm", "kernel": "k_tx (ragged)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes}}), flush=True)


def per_rank(x, dist, backend, dev, world):
    """x of every rank, in rank order (a list of floats on every rank)."""
    if not dist:
        return [float(x)]
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(q.item()) for q in parts]


def ranks_record(a, dist, backend, world, kernel_ms_by_rank):
    """What lets a reader of the line see that N ranks ran: the process group's own size, how the ranks
    were started and every rank's mean kernel time over the timed steps."""
    return {"world_size": dist.get_world_size() if dist else 1, "backend": ("rccl" if backend == "nccl" else backend) if dist else None,
            "gpus_requested": a.gpus,
            "started_by": "single process" if world == 1 else (
                "bench.py (one child process per rank)" if os.environ.get("BENCH_SELF_STARTED") else "external launcher"),
            "kernel_ms_per_rank": [round(v, 4) for v in kernel_ms_by_rank]}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; > 1 without a launcher's WORLD_SIZE: this process starts them")
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 600: 4.6 s of the headline kernel, long enough for a sampling monitor to see "
                         "and for the clocks to settle; 20 with --gather 1, whose steps are link-bound)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--packets", type=int, default=1_000_000, help="packets per GPU per step")
    ap.add_argument("--fused", type=int, default=1, help="1: single fused kernel (ofdmtx_tx), 0: two kernels")
    ap.add_argument("--stage-steps", type=int, default=3, help="untimed extra passes of the two-kernel path")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="packets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--workload", choices=["ftw", "ragged", "rawofdm"], default="ftw",
                    help="ftw: the headline (802.11a 54 Mbit/s packets); ragged: mixed rates and lengths; "
                         "rawofdm: the variable-size modulator")
    ap.add_argument("--ragged-packets", type=int, default=400_000, help="packets per GPU per step (ragged)")
    ap.add_argument("--raw-n", type=int, default=256)
    ap.add_argument("--raw-cp", type=int, default=64)
    ap.add_argument("--raw-symbols", type=int, default=1 << 20, help="data symbols per GPU per step (rawofdm)")
    ap.add_argument("--placement", choices=["spread", "probe", "first"], default="spread",
                    help="output buffer: pieces alternating between the parts of the device memory, the fastest of a "
                         "few ordinary allocations, or the first allocation as it comes")
    ap.add_argument("--placement-candidates", type=int, default=5)
    ap.add_argument("--inputs-spread", type=int, default=1,
                    help="1: the workload's input lies in the tail of the spread output buffer (pieces of all parts); "
                         "0: where the allocator put it")
    ap.add_argument("--placement-compare", type=int, default=1,
                    help="0: no comparison launch into an ordinary allocation (profiler runs: keeps the kernel's average clean)")
    ap.add_argument("--gather", type=int, default=0,
                    help="1: every step also collects all shards' IQ on rank 0 (config 5: RCCL send/recv, chunk k in "
                         "flight under the kernels of chunk k + 1); --packets is cut so that N shards fit the root")
    ap.add_argument("--gather-chunks", type=int, default=0,
                    help="chunks a shard is produced and sent in; 0: chunks of about 12 288 packets (0.59 GB of IQ), whose "
                         "k_tx launch ends within about 0.1 ms, so that the transfer kernels of the exchange are never "
                         "parked behind a long-running persistent grid")
    a = ap.parse_args(argv)
    if a.gpus < 1:
        ap.error("--gpus must be at least 1")
    if a.steps is None:
        a.steps = 20 if a.gather else 600
    return a


def start_ranks(n_ranks, argv):
    """Parent of a self-started run: N fresh children of this file, one per GPU, each with the rank
    environment a launcher would set.  Nothing here touches a GPU (no torch, no library), and no
    process image is replaced.  Rank 0's JSON line is relayed on stdout; everything else the ranks print
    goes to stderr.  Returns the exit status: 0 when every rank ended with 0, else the first failure's
    (after which the remaining ranks are terminated by PID)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, WORLD_SIZE=str(n_ranks), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_STARTED="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def relay():
        # stdout carries the JSON line and nothing else; whatever a library prints besides it
        # (gloo's connection notes, for one) goes to stderr
        for line in procs[0].stdout:
            dest = sys.stdout if line.startswith("{") else sys.stderr
            dest.write(line)
            dest.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    status = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            status = bad[0] if bad[0] > 0 else 128 - bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            sys.stderr.write("bench.py: a rank ended with status %s; exit codes %s\n" % (bad[0], [p.poll() for p in procs]))
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.1)
    t.join(timeout=5)
    return status


def main(a):
    load_runtime()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then
    # share devices); the real runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world != a.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n" % (a.gpus, world, world))
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    elif local >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d needs GPU %d, this node shows %d (BENCH_BACKEND=gloo rehearses more ranks than GPUs)"
                 % (rank, local, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    ctx = TxContext(local)
    if a.workload != "ftw":
        (run_rawofdm if a.workload == "rawofdm" else run_ragged)(a, ctx, dev, world, rank, dist, backend)
        if dist:
            dist.destroy_process_group()
        return
    n = a.packets
    info = _lib.packet_info(REGIME, PAYLOAD)
    gather = bool(a.gather) and world > 1
    if gather:
        # the root holds all N shards (48 000 B per packet): at most 60 % of the smaller of HBM size and 288 GB
        hbm = min(torch.cuda.get_device_properties(dev).total_memory, 288 * 10 ** 9)
        n = min(n, int(0.6 * hbm / (world * info.n_rows * 640)))
    # packets are sharded by index: rank r holds packets [r*n, (r+1)*n); the stream-wide pilot
    # counter of the reference continues across shards
    pilot_start = (rank * n * (info.n_sym + 1)) % 127
    plan = ctx.plan_uniform(n, REGIME, PAYLOAD, pilot_start=pilot_start)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)
    payloads = torch.randint(0, 256, (n * PAYLOAD,), dtype=torch.uint8, device=dev, generator=gen)
    msg = plan.alloc_msg() if not gather else None
    iq, placement = None, None
    if not gather:
        iq, placement, payloads = alloc_output(ctx, plan.n_samples, a.placement, lambda b: plan.tx(payloads, out=b, work=msg),
                                               a.placement_candidates, bool(a.placement_compare), inputs=payloads,
                                               move_inputs=bool(a.inputs_spread))
    if gather:
        from ofdm_amd.dist import chunk_bounds, gather_chunked
        spans = chunk_bounds(n, a.gather_chunks if a.gather_chunks > 0 else max(1, -(-n // 12288)))
        chunk_plans = [ctx.plan_uniform(hi - lo, REGIME, PAYLOAD,
                                        pilot_start=(pilot_start + lo * (info.n_sym + 1)) % 127) for lo, hi in spans]
        gathered = torch.empty(world * plan.n_samples, dtype=torch.complex64, device=dev) if rank == 0 else None

        def produce(k, buf):
            chunk_plans[k].tx(payloads[spans[k][0] * PAYLOAD:], out=buf)

    def step(events=None):
        """One pass of the transmit chain over the batch.  events (start, mid, end) are recorded on
        the launch stream around the kernels."""
        if events is not None:
            events[0].record()
        if gather:
            gather_chunked(produce, [p.n_samples for p in chunk_plans], dst=0, out=gathered, device=dev)
            if events is not None:
                events[1].record()
        elif a.fused:
            plan.tx(payloads, out=iq, work=msg)          # k_tx: payload -> IQ in one kernel
            if events is not None:
                events[1].record()
        else:
            plan.encode_bits(payloads, out=msg)          # k_encode_bits
            if events is not None:
                events[1].record()
            plan.modulate(msg, out=iq)                   # k_tx, message input
        if events is not None:
            events[2].record()

    dt, evs = timed_steps(step, a, dist, backend, dev)

    first_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    second_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    by_rank = per_rank(first_ms if (a.fused or gather) else second_ms, dist, backend, dev, world)

    # the two-kernel path, outside the timed region, for the per-stage numbers
    stage = None
    if a.fused and rank == 0 and a.stage_steps > 0 and not gather:
        plan.encode_bits(payloads, out=msg)
        plan.modulate(msg, out=iq)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.stage_steps)]
        for e in ev:
            e[0].record()
            plan.encode_bits(payloads, out=msg)
            e[1].record()
            plan.modulate(msg, out=iq)
            e[2].record()
        torch.cuda.synchronize()
        stage = (float(np.mean([e[0].elapsed_time(e[1]) for e in ev])),
                 float(np.mean([e[1].elapsed_time(e[2]) for e in ev])))

    # the fused map + IFFT + CP stage alone: mapper messages -> symbol rows, no preamble / gap rows
    # (2058 B read + 58 x 640 B written per packet), outside the timed region
    stage_only = None
    if a.fused and rank == 0 and a.stage_steps > 0 and not gather:
        bare = ctx.plan_uniform(n, REGIME, PAYLOAD, pilot_start=pilot_start, flags=0)
        bare.modulate(msg, out=iq)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(a.stage_steps)]
        for e in ev:
            e[0].record()
            bare.modulate(msg, out=iq)
            e[1].record()
        torch.cuda.synchronize()
        ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        sb = n * (info.msg_len + (info.n_sym + 1) * 640)
        stage_only = {"kernel": "k_tx (message input, flags = 0)", "ms": ms, "bytes_per_packet": info.msg_len + (info.n_sym + 1) * 640,
                      "GBps": sb / (ms * 1e-3) / 1e9, "frac_of_peak": sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "Msamples_per_s": n * (info.n_sym + 1) * 80 / (ms * 1e-3) / 1e6}

    if rank == 0:
        samples_per_step = world * plan.n_samples
        value = samples_per_step * a.steps / dt / 1e6
        if gather:
            kernel, kernel_ms = "k_tx in chunks + gather to rank 0 (overlapped)", first_ms
            alg_bytes = n * (PAYLOAD + info.n_rows * 640)
            kernels = {"k_tx_chunks_and_gather": first_ms}
        elif a.fused:
            kernel, kernel_ms = "k_tx", first_ms
            alg_bytes = n * (PAYLOAD + info.n_rows * 640)          # payload read + IQ rows written
            kernels = {"k_tx": first_ms}
        else:
            kernel, kernel_ms = "k_tx_msg", second_ms
            alg_bytes = n * (info.msg_len + info.n_rows * 640)     # mapper message read + IQ rows written
            kernels = {"k_encode_bits": first_ms, "k_tx_msg": second_ms}
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "baseband IQ Msamples/s (node), 802.11a 54Mbps 1500B packets",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "802.11a 54 Mbps (64-QAM r3/4), %d x 1500-byte packets per GPU, payload -> "
                                   "full PPDU IQ (4 preamble rows + SIGNAL + %d DATA symbols + 13 zero rows)"
                                   % (n, info.n_sym),
                       "packets_per_gpu": n, "samples_per_packet": info.n_rows * 80,
                       "parallelism": "packets sharded by index over %d GPU(s), %s" % (
                           world, "IQ gathered on rank 0 every step in %d overlapped chunks" % len(spans) if gather else "no collective")},
            "iq_placement": placement,
            "kernel_ms": kernels,
            "ranks": ranks_record(a, dist, backend, world, by_rank),
            # spread of the dominant kernel over the timed steps (first, fastest, slowest)
            "kernel_ms_steps": {"first": float(evs[0][0].elapsed_time(evs[0][1])),
                                "min": float(min(e[0].elapsed_time(e[1]) for e in evs)),
                                "max": float(max(e[0].elapsed_time(e[1]) for e in evs))},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic("%s_54M_1500B_%dpk" % (kernel, n)),
                         "traffic_source": "profiles/traffic.json (PMC passes of an earlier run of this kernel, "
                                           "not collected during this run)",
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if gather:
            # what arrives at the root per step: every peer's shard, each over its own xGMI link
            inbound = (world - 1) * n * info.n_rows * 640
            link = 153.6                                            # GB/s per xGMI link and direction (spec)
            out["gather"] = {"chunks": len(spans), "packets_per_chunk": spans[0][1] - spans[0][0],
                             "root_inbound_GBps": inbound * a.steps / dt / 1e9,
                             "root_inbound_bound_GBps": (world - 1) * link,
                             "frac_of_bound": inbound * a.steps / dt / 1e9 / ((world - 1) * link),
                             "note": "bound = one xGMI link per peer into the root (spec rate); a rehearsal over gloo "
                                     "moves the shards through host memory and says nothing about the links"}
        if stage is not None:
            mb = n * (info.msg_len + info.n_rows * 640)
            # map + IFFT + CP stage on its own: mapper messages (2058 B/packet) -> IQ
            out["two_kernel_path"] = {"k_encode_bits_ms": stage[0], "k_tx_msg_ms": stage[1],
                                      "k_tx_msg_GBps": mb / (stage[1] * 1e-3) / 1e9,
                                      "k_tx_msg_frac_of_peak": mb / (stage[1] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if stage_only is not None:
            out["map_ifft_cp_only"] = stage_only
        if world == 1 and a.cpu_sample > 0:
            threads = host_threads()
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample, threads)
            out["cpu_baseline"]["host_cpu_count"] = os.cpu_count()
            out["cpu_baseline"]["config1_py_refshaped"] = py_refshaped_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(start_ranks(args.gpus, sys.argv[1:]))
    main(args)
