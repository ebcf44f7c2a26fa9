"""ofdmtx_tx_ragged_static -- the entry point that allocates nothing -- against the existing pair
ofdmtx_plan_ragged_device + ofdmtx_tx, through the C ABI: bit-identical IQ on two calls in a row,
the second with a smaller batch (the static work area keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from ofdm_amd import _lib

pytestmark = pytest.mark.gpu


def batch(rng, n, dev):
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 1600, n).astype(np.int32)
    lens[: min(n, 3)] = [0, 1, 3140][: min(n, 3)]
    payloads = rng.integers(0, 256, max(int(lens.sum()), 1), dtype=np.uint8)
    return (torch.from_numpy(regimes).to(dev), torch.from_numpy(lens).to(dev), torch.from_numpy(payloads).to(dev))


def test_static_entry_point_matches_plan_and_tx(txctx):
    dev = txctx.device
    rng = np.random.default_rng(77)
    out = torch.empty(48_000_000, dtype=torch.complex64, device=dev)
    for n, pilot_start, flags, amp in ((3000, 17, _lib.F_PREAMBLE | _lib.F_ZEROGAP, 0.3),
                                       (700, 3, _lib.F_PREAMBLE | _lib.F_ZEROGAP | _lib.F_SWAPIQ, 1.0),
                                       (1, 0, 0, 0.5)):
        regimes, lens, payloads = batch(rng, n, dev)
        plan = txctx.plan_ragged(regimes, lens, pilot_start=pilot_start, flags=flags, amp=amp)      # device-built plan
        want = plan.tx(payloads)
        out.fill_(float("nan"))
        torch.cuda.synchronize()
        # what the DRIVER says is free on the device (hipMemGetInfo), not what torch's allocator has handed
        # out: a hipMalloc or hipFree inside the library would show here.  (The first launch of a kernel in
        # a process loads its code object, so the first round is not held to it.)
        free_before, _ = torch.cuda.mem_get_info(dev)
        got_n = txctx.tx_ragged_static(regimes, lens, payloads, out, pilot_start=pilot_start, flags=flags, amp=amp)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(dev)
        if n != 3000:
            assert free_after == free_before, "device memory changed by %d bytes across the call" % (free_before - free_after)
        assert got_n == plan.n_samples
        assert torch.equal(out[:got_n].view(torch.float32), want.view(torch.float32))
        assert bool(torch.isnan(out[got_n:got_n + 4096].real).all())          # nothing written past the batch


def test_static_entry_point_limits(txctx):
    dev = txctx.device
    rng = np.random.default_rng(78)
    regimes, lens, payloads = batch(rng, 50, dev)
    small = torch.empty(1000, dtype=torch.complex64, device=dev)
    with pytest.raises(ValueError):
        txctx.tx_ragged_static(regimes, lens, payloads, small)
    n = _lib.STATIC_MAX_PKTS + 1
    with pytest.raises(_lib.OfdmTxError) as e:
        txctx.tx_ragged_static(torch.ones(n, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                               payloads, small)
    assert e.value.code == _lib.ETOOBIG
    bad = regimes.clone()
    bad[7] = 9
    big = torch.empty(2_000_000, dtype=torch.complex64, device=dev)
    with pytest.raises(_lib.OfdmTxError) as e:
        txctx.tx_ragged_static(bad, lens, payloads, big)
    assert e.value.code == _lib.EREGIME
    # and the area is usable again after a refused call
    got = txctx.tx_ragged_static(regimes, lens, payloads, big)
    want = txctx.plan_ragged(regimes, lens).tx(payloads)
    torch.cuda.synchronize()
    assert torch.equal(big[:got].view(torch.float32), want.view(torch.float32))


def test_static_entry_point_calls_on_two_streams_are_ordered(txctx):
    """All calls on a GPU share one work area.  A second call on another stream, issued while the first
    call's kernel is still running (and through a second context), is ordered behind it by the library:
    both batches come out right."""
    from ofdm_amd.batch import TxContext
    dev = txctx.device
    rng = np.random.default_rng(79)
    other = TxContext(dev.index)
    side = torch.cuda.Stream(device=dev)
    big = batch(rng, 60000, dev)
    small = batch(rng, 500, dev)
    want_big = txctx.plan_ragged(big[0], big[1], pilot_start=5).tx(big[2])
    want_small = txctx.plan_ragged(small[0], small[1], pilot_start=9).tx(small[2])
    out_big = torch.zeros(want_big.numel(), dtype=torch.complex64, device=dev)
    out_small = torch.zeros(want_small.numel(), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        n_big = txctx.tx_ragged_static(big[0], big[1], big[2], out_big, pilot_start=5)        # ~1 ms of kernel
        with torch.cuda.stream(side):
            n_small = other.tx_ragged_static(small[0], small[1], small[2], out_small, pilot_start=9)
        torch.cuda.synchronize()
        assert n_big == want_big.numel() and n_small == want_small.numel()
        assert torch.equal(out_big.view(torch.float32), want_big.view(torch.float32))
        assert torch.equal(out_small.view(torch.float32), want_small.view(torch.float32))
        out_big.zero_()
        out_small.zero_()
    other.close()
