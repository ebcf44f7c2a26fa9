"""Packet-index sharding over the GPUs of one node (one process per GPU).

Packets are independent units (fixed scrambler seed, encoder flushed per packet, interleaver
per symbol), so a batch is split into contiguous packet ranges and every rank runs the
kernels on its own range: no collective on the data path.  The only stream-wide state of
the reference is the pilot block's running symbol counter
(ftw_80211_ofdm_tx/src/ftw/ftw_ofdm_pilot_cc.cc:82,159); a shard continues it by starting
at the number of symbols in all earlier packets.  `gather_iq` is the optional exchange step
that collects the shards' IQ into one buffer on a root rank (RCCL over xGMI when the tensors
live on GPUs, gloo on CPU tensors).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_bounds(n_pkts, world, rank):
    """Contiguous, balanced packet range [lo, hi) of `rank`."""
    lo = n_pkts * rank // world
    hi = n_pkts * (rank + 1) // world
    return lo, hi


def shard_bounds_by_rows(rows_per_pkt, world, rank):
    """Ragged batches: contiguous ranges balanced by output rows (80-sample rows) rather than
    by packet count.  rows_per_pkt: array with each packet's row count."""
    rows = np.asarray(rows_per_pkt, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(rows)])
    total = cum[-1]
    lo = int(np.searchsorted(cum, total * rank / world, side="left"))
    hi = int(np.searchsorted(cum, total * (rank + 1) / world, side="left"))
    if rank == world - 1:
        hi = len(rows)
    return min(lo, len(rows)), min(hi, len(rows))


def pilot_start_of_shard(n_sym_before, pilot_start=0):
    """Polarity index of the first symbol of a shard whose earlier packets hold
    `n_sym_before` = sum(N_sym + 1) OFDM symbols (SIGNAL included)."""
    return int((pilot_start + int(n_sym_before)) % 127)


def gather_iq(iq, dst=0, group=None):
    """Collect every rank's IQ tensor (1-D complex64, lengths may differ) on rank `dst` in rank
    order.  Returns the concatenated tensor on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([iq.numel()], dtype=torch.int64, device=iq.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    flat = torch.view_as_real(iq).reshape(-1)
    if rank == dst:
        out = torch.empty(sum(sizes), dtype=torch.complex64, device=iq.device)
        out_r = torch.view_as_real(out).reshape(-1)
        reqs = []
        off = 0
        for r in range(world):
            seg = out_r[2 * off:2 * (off + sizes[r])]
            if r == rank:
                seg.copy_(flat)
            elif sizes[r]:
                reqs.append(dist.irecv(seg, src=r, group=group))
            off += sizes[r]
        for q in reqs:
            q.wait()
        return out
    if iq.numel():
        dist.send(flat, dst=dst, group=group)
    return None


_side_streams = {}


def _side_stream(device):
    """The copy / transfer stream of `device` for gather_chunked, created once per device."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def chunk_bounds(n_pkts, n_chunks):
    """Packet ranges [lo, hi) of the chunks a shard is produced and sent in (n_chunks >= 1, balanced)."""
    n_chunks = max(1, min(int(n_chunks), max(int(n_pkts), 1)))
    return [(n_pkts * k // n_chunks, n_pkts * (k + 1) // n_chunks) for k in range(n_chunks)]


def gather_chunked(produce, chunk_samples, dst=0, group=None, out=None, device=None):
    """Produce this rank's IQ in chunks and collect every rank's chunks on rank `dst`, the transfer of
    chunk k running under the kernels of chunk k + 1.

    produce(k, buf): enqueue on the CURRENT stream whatever fills `buf` (1-D complex64 tensor of
    chunk_samples[k] elements on `device`) with chunk k.  chunk_samples: this rank's chunk sizes
    (ranks may differ).  Returns on `dst` a 1-D complex64 tensor on `device`: all ranks' chunks, rank
    after rank, chunk after chunk; None on the other ranks.

    Root: its own chunks are produced straight into their place in the result, the peers' chunks are
    received into place -- one grouped receive per chunk index, so the peers' transfers (each on its
    own xGMI link to the root) run side by side.  Peers: two chunk buffers; chunk k is sent from a side
    stream once its kernels have finished while chunk k + 1 is being produced into the other buffer.
    So a peer never holds more than two chunks, and the root exactly the result.
    With a backend that cannot move device memory (gloo: the rehearsal of this path on one GPU or on
    the CPU) device chunks travel through page-locked host buffers; with `device` = cpu there are no
    streams at all and the same protocol runs on host tensors."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    direct = dist.get_backend(group) == "nccl"           # the backend moves device memory itself
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    gpu = device.type == "cuda"
    if direct and not gpu:
        raise ValueError("the nccl backend needs device tensors")
    sizes_local = [int(c) for c in chunk_samples]
    sizes_all = [None] * world                           # every rank's chunk sizes (a few integers, once per call)
    dist.all_gather_object(sizes_all, sizes_local, group=group)
    compute = torch.cuda.current_stream(device) if gpu else None
    side = _side_stream(device) if gpu else None
    if gpu:
        # whatever the compute stream still does with `out` or with recycled memory (a consumer of the
        # previous call's result, kernels of a tensor the caching allocator has just handed back) comes
        # first: nothing is received or sent from the side stream before it
        side.wait_stream(compute)

    def host_buffer(n_floats):
        return torch.empty(n_floats, dtype=torch.float32, pin_memory=gpu)

    if rank == dst:
        starts, total = [], 0
        for r in range(world):
            row = []
            for c in sizes_all[r]:
                row.append(total)
                total += c
            starts.append(row)
        if out is None:
            out = torch.empty(total, dtype=torch.complex64, device=device)
        elif out.numel() < total or out.dtype != torch.complex64 or not out.is_contiguous() or out.device != device:
            raise ValueError("gather buffer must be contiguous complex64 on %s with >= %d samples" % (device, total))
        flat = torch.view_as_real(out).reshape(-1)
        piece = lambda r, k: flat[2 * starts[r][k]:2 * (starts[r][k] + sizes_all[r][k])]
        reqs, staged = [], []
        for k in range(max(len(c) for c in sizes_all)):
            # the receives of chunk k from every peer, posted before the root's own chunk k is
            # produced so that they proceed under its kernels
            ops = []
            for r in range(world):
                if r == dst or k >= len(sizes_all[r]) or sizes_all[r][k] == 0:
                    continue
                if direct or not gpu:
                    ops.append(dist.P2POp(dist.irecv, piece(r, k), r, group))
                else:
                    h = host_buffer(2 * sizes_all[r][k])
                    staged.append((h, r, k))
                    ops.append(dist.P2POp(dist.irecv, h, r, group))
            if ops and direct:
                with torch.cuda.stream(side):
                    reqs.extend(dist.batch_isend_irecv(ops))
            elif ops:
                reqs.extend(dist.irecv(op.tensor, src=op.peer, group=group) for op in ops)
            if k < len(sizes_local) and sizes_local[k]:
                produce(k, out[starts[dst][k]:starts[dst][k] + sizes_local[k]])
        for q in reqs:
            q.wait()
        for h, r, k in staged:
            piece(r, k).copy_(h, non_blocking=True)
        if gpu:
            compute.wait_stream(side)
        return out

    # ---- a peer: two chunk buffers, chunk k on its way while chunk k + 1 is produced
    biggest = max(sizes_local + [1])
    bufs = [torch.empty(biggest, dtype=torch.complex64, device=device) for _ in range(2)]
    hosts = [host_buffer(2 * biggest) for _ in range(2)] if (gpu and not direct) else None
    pending = [None, None]               # per buffer: (request, event) of the send that still reads it
    for k, n in enumerate(sizes_local):
        if n == 0:
            continue
        b = k & 1
        if pending[b] is not None:       # the buffer's previous chunk must have left
            req, ev = pending[b]
            req.wait()
            if ev is not None:
                compute.wait_event(ev)
        buf = bufs[b][:n]
        produce(k, buf)
        flat = torch.view_as_real(buf).reshape(-1)
        if not gpu:
            pending[b] = (dist.isend(flat, dst=dst, group=group), None)
            continue
        done = torch.cuda.Event()
        done.record(compute)
        with torch.cuda.stream(side):
            side.wait_event(done)
            if direct:
                req = dist.isend(flat, dst=dst, group=group)
                ev = torch.cuda.Event()
                ev.record(side)
            else:
                h = hosts[b][:2 * n]
                h.copy_(flat, non_blocking=True)
                side.synchronize()       # gloo reads host memory from the CPU
                req, ev = dist.isend(h, dst=dst, group=group), None
        pending[b] = (req, ev)
    for p in pending:
        if p is not None:
            p[0].wait()
    if gpu:
        compute.wait_stream(side)
    return None


class Comm(object):
    """The exchange step at the C ABI (include/ofdmtx.h, ofdmtx_comm_* / ofdmtx_gather): one RCCL
    communicator per rank, for hosts that do not run torch.distributed.  Rank 0 calls
    Comm.unique_id() and hands the 128 bytes to the other ranks; every rank then constructs
    Comm(ctx, rank, world, uid) (collective)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().ofdmtx_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, rank, world, uid):
        if len(uid) != 128:
            raise ValueError("the communicator id has 128 bytes")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = C.c_void_p()
        _lib.check(ctx._lib.ofdmtx_comm_create(ctx._h, self.rank, self.world, C.c_char_p(uid), C.byref(h)))
        self._h = h

    def gather(self, src, sample_off=None, out=None, root=0, stream=None):
        """src: this rank's complex64 samples on the context's device.  On `root`: sample_off = the
        world + 1 ascending sample offsets of the ranks' shards in `out` (allocated when None);
        returns `out` there, None elsewhere.  Enqueued on `stream` (default: the current one)."""
        dev = self.ctx.device
        if src.dtype != torch.complex64 or src.device != dev or not src.is_contiguous():
            raise ValueError("src must be a contiguous complex64 tensor on %s" % dev)
        st = C.c_void_p((stream or torch.cuda.current_stream(dev)).cuda_stream)
        if self.rank != root:
            _lib.check(self.ctx._lib.ofdmtx_gather(self._h, C.c_void_p(src.data_ptr()), src.numel(), None, None,
                                                   int(root), st))
            return None
        off = np.ascontiguousarray(sample_off, dtype=np.uint64)
        if off.shape != (self.world + 1,):
            raise ValueError("sample_off needs world + 1 entries")
        if out is None:
            out = torch.empty(int(off[-1]), dtype=torch.complex64, device=dev)
        elif out.numel() < int(off[-1]) or out.dtype != torch.complex64 or out.device != dev or not out.is_contiguous():
            raise ValueError("gather buffer must be contiguous complex64 on %s with >= %d samples" % (dev, int(off[-1])))
        _lib.check(self.ctx._lib.ofdmtx_gather(self._h, C.c_void_p(src.data_ptr()), src.numel(),
                                               C.c_void_p(out.data_ptr()), off.ctypes.data, int(root), st))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.ofdmtx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
