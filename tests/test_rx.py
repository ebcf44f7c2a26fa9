"""Loop-back verifier (ofdmtx_rx_demod / ofdmtx_rx_decode): a receiver written from the 802.11a
standard decodes, on the GPU, what the transmit kernels (and the CPU oracle) produce.

Clean loop-back: every status word 0, the payload bytes come back exactly, and the hard coded bits
equal the transmitter's `coded` tap bit for bit.  Damaged frames raise the matching status bits."""
import numpy as np
import pytest
import torch

from ofdm_amd import _lib
from oracle import ftw_oracle as o

pytestmark = pytest.mark.gpu


def coded_equal(plan, got, want):
    lay = plan.layout()
    g, w = got.cpu().numpy(), want.cpu().numpy()
    for i in range(plan.n_pkts):
        mo, ml = int(lay["msg_off"][i]), int(lay["msg_len"][i])
        if g[mo:mo + ml].tobytes() != w[mo:mo + ml].tobytes():
            return False
    return True


@pytest.mark.parametrize("regime", range(1, 9))
def test_loopback_all_rates(txctx, regime):
    rng = np.random.default_rng(40 + regime)
    for ln in (0, 1, 37, 1500, 3140):
        n = 5
        pay = torch.from_numpy(rng.integers(0, 256, n * ln, dtype=np.uint8)).to(txctx.device)
        plan = txctx.plan_uniform(n, regime, ln, pilot_start=11)
        iq = plan.tx(pay)
        coded, status = plan.rx_demod(iq)
        assert int(status.abs().max()) == 0
        assert coded_equal(plan, coded, plan.encode_bits(pay, taps=True)["coded"])
        got, status = plan.rx_decode(coded, status)
        assert status.cpu().tolist() == [0] * n, (regime, ln)
        assert torch.equal(got, pay[:plan.payload_bytes])


@pytest.mark.parametrize("flags,pilot_mode", [
    (0, _lib.PILOT_PER_PACKET), (_lib.F_PREAMBLE | _lib.F_ZEROGAP | _lib.F_SWAPIQ, _lib.PILOT_REF_STREAM),
    (_lib.F_PREAMBLE | _lib.F_PNC_A | _lib.F_ZEROGAP, _lib.PILOT_PNC_A), (_lib.F_PREAMBLE | _lib.F_PNC_B, _lib.PILOT_PNC_B),
    (_lib.F_ZEROGAP, _lib.PILOT_PNC_ALL)])
def test_loopback_ragged_framing_and_pilot_modes(txctx, flags, pilot_mode):
    rng = np.random.default_rng(50 + flags)
    n = 300
    regs = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 700, n).astype(np.uint32)
    lens[:3] = [0, 2304, 3140]
    pay = torch.from_numpy(rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)).to(txctx.device)
    plan = txctx.plan_ragged(regs, lens, pilot_mode=pilot_mode, pilot_start=99, flags=flags, amp=0.7)
    got, status = plan.loopback(plan.tx(pay))
    assert int(status.abs().max()) == 0
    assert torch.equal(got, pay)
    # the message-input path gives the same frames
    got, status = plan.loopback(plan.modulate(plan.encode_bits(pay)))
    assert int(status.abs().max()) == 0 and torch.equal(got, pay)


def test_receiver_decodes_oracle_frames(txctx, c_oracle):
    """Frames built by the CPU oracle, uploaded: the receiver is checked without the GPU transmitter."""
    rng = np.random.default_rng(60)
    regs = [1, 2, 3, 4, 5, 6, 7, 8, 8, 1]
    lens = [10, 200, 33, 0, 512, 100, 1500, 77, 3140, 1]
    pays = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    plan = txctx.plan_ragged(np.array(regs, np.uint8), np.array(lens, np.uint32), pilot_start=5, amp=0.3)
    lay = plan.layout()
    iq = np.zeros(plan.n_samples, np.complex64)
    for i, (r, p) in enumerate(zip(regs, pays)):
        enc = c_oracle.encode_bits(p.tobytes(), r)
        one = c_oracle.modulate(enc["interleaved"], r, int(lay["pilot_base"][i]), 0.3, 3)["iq"]
        a = int(lay["sample_off"][i])
        iq[a:a + one.size] = one
    got, status = plan.loopback(torch.from_numpy(iq).to(txctx.device))
    assert status.cpu().tolist() == [0] * len(regs)
    assert got.cpu().numpy().tobytes() == b"".join(p.tobytes() for p in pays)


def test_damage_is_reported(txctx):
    rng = np.random.default_rng(61)
    n, ln, regime = 8, 400, 6
    pay = torch.from_numpy(rng.integers(0, 256, n * ln, dtype=np.uint8)).to(txctx.device)
    plan = txctx.plan_uniform(n, regime, ln, pilot_start=3)
    iq = plan.tx(pay).clone()
    rows = iq.view(n, -1, 80)
    rows[1, 4 + 1 + 5] = 0                      # a DATA symbol wiped out: pilots gone, bits wrong
    rows[2, 4] *= -1                            # SIGNAL symbol inverted
    rows[3, 4 + 1 + 2, 16:] += 0.05             # energy on the DC carrier (a constant over the FFT window)
    coded, status = plan.rx_demod(iq)
    st = status.cpu().numpy()
    assert st[0] == 0 and all(st[4:] == 0)
    assert st[1] & _lib.RX_PILOTS and st[2] & _lib.RX_PILOTS and st[3] & _lib.RX_NULLS
    c = coded.clone()
    lay = plan.layout()
    c[int(lay["msg_off"][5]) + 100] ^= 0x10     # one coded bit flipped in packet 5
    got, status = plan.rx_decode(c, status.clone())
    st = status.cpu().numpy()
    assert st[0] == 0 and st[4] == 0 and st[6] == 0 and st[7] == 0
    assert st[1] & (_lib.RX_CODE | _lib.RX_FCS)
    assert st[2] & _lib.RX_SIGNAL
    assert st[5] & (_lib.RX_CODE | _lib.RX_FCS)
    g = got.view(n, ln)
    assert torch.equal(g[0], pay.view(n, ln)[0]) and torch.equal(g[7], pay.view(n, ln)[7])
    # a plan that expects another length or rate says so
    other = txctx.plan_uniform(n, regime, ln, pilot_start=3, amp=0.5)      # wrong amplitude: every decision fails
    _, status = other.rx_demod(plan.tx(pay))
    assert bool((status & _lib.RX_PILOTS).bool().all())


def test_custom_constellation_loops_back(txctx):
    """The de-mapper decides against the context's constellation table, whatever its points."""
    from ofdm_amd.batch import TxContext
    ctx = TxContext(0)
    try:
        pts = np.array([1 + 1j, -1 + 1j, 1 - 1j, -1 - 1j], np.complex64) * np.float32(0.5)
        ctx.set_constellation(2, pts)
        rng = np.random.default_rng(62)
        pay = torch.from_numpy(rng.integers(0, 256, 4 * 90, dtype=np.uint8)).to(ctx.device)
        plan = ctx.plan_uniform(4, 4, 90)
        got, status = plan.loopback(plan.tx(pay))
        assert int(status.abs().max()) == 0 and torch.equal(got, pay)
        plan.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("kind", ["rotated", "product"])
@pytest.mark.parametrize("nbits,regime", [(4, 6), (6, 8)])
def test_custom_qam_tables_loop_back(kind, nbits, regime):
    """16-QAM / 64-QAM tables of the caller's own: one whose points are NOT (I[first half of the bits],
    Q[second half]) -- the standard table turned by 30 degrees, searched point by point -- and one that is such
    a product with unevenly spaced levels in no particular order (searched axis by axis)."""
    from ofdm_amd.batch import TxContext
    ctx = TxContext(0)
    try:
        h = nbits // 2
        if kind == "rotated":
            std = o.tables().constellation["qam16" if nbits == 4 else "qam64"].astype(np.complex64)
            pts = (std * np.exp(1j * np.pi / 6)).astype(np.complex64)
        else:
            lv = np.random.default_rng(7).permutation(np.cumsum(np.linspace(0.11, 0.23, 1 << h)) - 0.6).astype(np.float32)
            idx = np.arange(1 << nbits)
            pts = (lv[idx >> h] + 1j * lv[::-1][idx & ((1 << h) - 1)]).astype(np.complex64)
        ctx.set_constellation(nbits, pts)
        rng = np.random.default_rng(63)
        lens = rng.integers(1, 700, 37).astype(np.uint32)
        pay = torch.from_numpy(rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)).to(ctx.device)
        plan = ctx.plan_ragged(np.full(37, regime, np.uint8), lens)
        got, status = plan.loopback(plan.tx(pay))
        assert int(status.abs().max()) == 0 and torch.equal(got, pay)
        plan.close()
    finally:
        ctx.close()
