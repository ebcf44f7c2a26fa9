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
import numpy as np
import torch
import torch.distributed as dist


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
