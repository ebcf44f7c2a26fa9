"""GPU parity: HIP kernels (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact for PLCP / scrambled / coded / interleaved bytes and for mapped symbols;
IQ within 1e-5 relative L2 per OFDM symbol (row of 80 samples) against the float64 oracle.
"""
import numpy as np
import pytest
import torch

from ofdm_amd import _lib

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-5   # per 80-sample row, north-star tolerance for fp32 IQ


def oracle_batch(c_oracle, payloads, lens, regimes, pilot_base, flags, amp, symbol_time=4.0):
    """Per-packet oracle results: list of dicts (bytes taps, mapped, iq64)."""
    out = []
    off = 0
    for i, (n, r) in enumerate(zip(lens, regimes)):
        pl = payloads[off:off + n].tobytes()
        off += n
        if flags & _lib.F_RAW_PSDU:                       # the packet's bytes are the whole PSDU
            from oracle import ftw_oracle
            enc = ftw_oracle.encode_bits_from_psdu(pl, str(int(r)))
        else:
            enc = c_oracle.encode_bits(pl, int(r), symbol_time)
        mod = c_oracle.modulate(enc["interleaved"], int(r), int(pilot_base[i]), amp, flags & ~_lib.F_RAW_PSDU)
        enc.update(mod)
        out.append(enc)
    return out


def rel_l2_rows(got, ref):
    got = got.reshape(-1, 80).astype(np.complex128)
    ref = ref.reshape(-1, 80)
    num = np.sqrt((np.abs(got - ref) ** 2).sum(axis=1))
    den = np.sqrt((np.abs(ref) ** 2).sum(axis=1))
    zero = den == 0
    assert np.all(num[zero] == 0), "non-zero samples where the oracle has an all-zero row"
    return np.where(zero, 0.0, num / np.where(zero, 1.0, den))


def check_plan(plan, payloads_np, lens, regimes, c_oracle, flags, amp, symbol_time=4.0):
    dev = plan.ctx.device
    payloads = torch.from_numpy(payloads_np).to(dev)
    taps = plan.encode_bits(payloads, taps=True)
    iq, mapped = plan.modulate(taps["interleaved"], want_mapped=True)
    iq2 = plan.tx(payloads)
    torch.cuda.synchronize()
    lay = plan.layout()
    ref = oracle_batch(c_oracle, payloads_np, lens, regimes, lay["pilot_base"], flags, amp, symbol_time)
    t = {k: v.cpu().numpy() for k, v in taps.items()}
    iq = iq.cpu().numpy()
    iq2 = iq2.cpu().numpy()
    mapped = mapped.cpu().numpy()
    assert np.array_equal(iq.view(np.uint64), iq2.view(np.uint64)), "tx() differs from encode_bits()+modulate()"
    worst = 0.0
    for i, r in enumerate(ref):
        po, mo = int(lay["plcp_off"][i]), int(lay["msg_off"][i])
        pl, ml = int(lay["plcp_len"][i]), int(lay["msg_len"][i])
        assert pl == len(r["plcp"]) and ml == len(r["interleaved"])
        assert t["plcp"][po:po + pl].tobytes() == r["plcp"], "plcp bytes, packet %d" % i
        assert t["scrambled"][po:po + pl].tobytes() == r["scrambled"], "scrambled bytes, packet %d" % i
        assert t["coded"][mo:mo + ml].tobytes() == r["coded"], "coded bytes, packet %d" % i
        assert t["interleaved"][mo:mo + ml].tobytes() == r["interleaved"], "interleaved bytes, packet %d" % i
        so, ns = int(lay["sym_off"][i]), int(lay["n_sym"][i])
        assert np.array_equal(mapped[so:so + ns + 1].view(np.uint64), r["mapped"].view(np.uint64)), \
            "mapped symbols, packet %d" % i
        a = int(lay["sample_off"][i])
        n = len(r["iq64"])
        e = rel_l2_rows(iq[a:a + n], r["iq64"])
        worst = max(worst, e.max())
        assert e.max() <= REL_L2_TOL, "IQ rel L2 %.3g in packet %d row %d" % (e.max(), i, e.argmax())
    assert sum(len(r["iq64"]) for r in ref) == plan.n_samples
    return worst


@pytest.mark.parametrize("regime", range(1, 9))
@pytest.mark.parametrize("payload_len", [0, 1, 10, 100, 1500])
def test_uniform_all_regimes(txctx, c_oracle, regime, payload_len):
    rng = np.random.default_rng(1000 * regime + payload_len)
    n = 5
    payloads = rng.integers(0, 256, n * payload_len, dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
    plan = txctx.plan_uniform(n, regime, payload_len, pilot_start=3, flags=flags, amp=0.3)
    check_plan(plan, payloads, [payload_len] * n, [regime] * n, c_oracle, flags, 0.3)


@pytest.mark.parametrize("flags", [0, _lib.F_PREAMBLE, _lib.F_ZEROGAP, _lib.F_PREAMBLE | _lib.F_ZEROGAP | _lib.F_SWAPIQ])
def test_framing_flags(txctx, c_oracle, flags):
    rng = np.random.default_rng(7 + flags)
    n, ln = 9, 333
    payloads = rng.integers(0, 256, n * ln, dtype=np.uint8)
    plan = txctx.plan_uniform(n, 6, ln, pilot_mode=_lib.PILOT_PER_PACKET, flags=flags, amp=1.0)
    check_plan(plan, payloads, [ln] * n, [6] * n, c_oracle, flags, 1.0)


def test_max_length(txctx, c_oracle):
    rng = np.random.default_rng(11)
    for regime in (1, 8):
        payloads = rng.integers(0, 256, 3140, dtype=np.uint8)
        plan = txctx.plan_uniform(1, regime, 3140)
        check_plan(plan, payloads, [3140], [regime], c_oracle, 3, 0.3)
    with pytest.raises(ValueError):
        txctx.plan_uniform(1, 8, 3141)


def test_ragged_mixed_rates(txctx, c_oracle):
    rng = np.random.default_rng(3)
    n = 200
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 2305, n).astype(np.uint32)
    lens[:4] = [0, 1, 2304, 3140]
    payloads = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
    plan = txctx.plan_ragged(regimes, lens, pilot_start=100, flags=flags, amp=0.3)
    check_plan(plan, payloads, lens.tolist(), regimes.tolist(), c_oracle, flags, 0.3)


@pytest.mark.parametrize("regime", range(1, 9))
def test_raw_psdu_uniform(txctx, c_oracle, regime):
    """Plans with OFDMTX_F_RAW_PSDU: the caller's bytes are MPDU + FCS; every tap and the IQ against the
    oracle's chain for the same PSDU, fused == two-kernel bitwise."""
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | _lib.F_RAW_PSDU
    for ln in (0, 1, 3, 4, 5, 100, 1528):
        rng = np.random.default_rng(50 * regime + ln)
        n = 4
        psdus = rng.integers(0, 256, n * ln, dtype=np.uint8)
        plan = txctx.plan_uniform(n, regime, ln, pilot_start=9, flags=flags, amp=0.3)
        check_plan(plan, psdus, [ln] * n, [regime] * n, c_oracle, flags, 0.3)


def test_raw_psdu_ragged_and_limits(txctx, c_oracle):
    rng = np.random.default_rng(17)
    n = 120
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 2400, n).astype(np.uint32)
    lens[:5] = [0, 1, 3168, 4, 2]
    psdus = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | _lib.F_RAW_PSDU
    plan = txctx.plan_ragged(regimes, lens, pilot_start=31, flags=flags, amp=0.3)
    check_plan(plan, psdus, lens.tolist(), regimes.tolist(), c_oracle, flags, 0.3)
    # the same plan built on the device
    dev = txctx.device
    plan_d = txctx.plan_ragged(torch.from_numpy(regimes).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev),
                               pilot_start=31, flags=flags, amp=0.3)
    check_plan(plan_d, psdus, lens.tolist(), regimes.tolist(), c_oracle, flags, 0.3)
    # a PSDU is limited like MPDU + FCS (examples/ftw_packet_utils.py:119-121); PNC roles are refused
    with pytest.raises(ValueError):
        txctx.plan_uniform(1, 8, 3169, flags=flags)
    with pytest.raises(_lib.OfdmTxError):
        txctx.plan_uniform(1, 8, 100, flags=flags | _lib.F_PNC_A)
    # an ordinary packet sent as a raw PSDU (own header, own CRC) gives the ordinary plan's samples, and
    # the loop-back verifier returns the whole PSDU with a clean status
    from oracle import ftw_oracle as o
    payload = rng.integers(0, 256, 321, dtype=np.uint8).tobytes()
    mpdu = o.get_info(payload, "8", 4.0)["packet"]
    whole = mpdu + o._bitrev8(o.ftw_crc32(mpdu).to_bytes(4, "big"))
    std = txctx.plan_uniform(1, 8, len(payload), amp=1.0)
    raw = txctx.plan_uniform(1, 8, len(whole), amp=1.0, flags=flags)
    iq_std = std.tx(torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev))
    iq_raw = raw.tx(torch.frombuffer(bytearray(whole), dtype=torch.uint8).to(dev))
    assert torch.equal(iq_std.view(torch.float32), iq_raw.view(torch.float32))
    back, status = raw.loopback(iq_raw)
    assert int(status.abs().max()) == 0 and back.cpu().numpy().tobytes() == whole


def test_stream_pilot_counter_matches_flowgraph_order(txctx):
    """REF_STREAM pilots: packet q starts at sum of (N_sym+1) of earlier packets, mod 127."""
    plan = txctx.plan_uniform(300, 8, 1500)
    pb = plan.layout()["pilot_base"]
    assert np.array_equal(pb, (np.arange(300) * 58) % 127)


@pytest.mark.parametrize("regime,n_sym", [(8, 63), (8, 64), (8, 65), (8, 117), (7, 64), (1, 64), (1, 65), (3, 128), (3, 129), (3, 130)])
def test_symbol_chunk_boundaries(txctx, c_oracle, regime, n_sym):
    """The fused kernel builds slot rows for 64 DATA symbols at a time: packets whose symbol
    count sits on either side of that boundary, uniform and inside a ragged batch."""
    n_dbps = {1: 24, 3: 48, 7: 192, 8: 216}[regime]
    payload_len = (n_sym * n_dbps - 54) // 8 - 24            # the longest payload with exactly n_sym symbols
    assert _lib.packet_info(regime, payload_len).n_sym == n_sym
    rng = np.random.default_rng(100 * regime + n_sym)
    n = 3
    payloads = rng.integers(0, 256, n * payload_len, dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
    plan = txctx.plan_uniform(n, regime, payload_len, pilot_start=100, flags=flags, amp=0.3)
    check_plan(plan, payloads, [payload_len] * n, [regime] * n, c_oracle, flags, 0.3)
    lens = [payload_len, 10, payload_len]
    regs = [regime, 5, regime]
    pay = rng.integers(0, 256, sum(lens), dtype=np.uint8)
    plan = txctx.plan_ragged(regs, lens, pilot_start=100, flags=flags, amp=0.3)
    check_plan(plan, pay, lens, regs, c_oracle, flags, 0.3)


def test_empty_plans(txctx):
    plan = txctx.plan_uniform(0, 8, 100)
    assert plan.n_samples == 0 and plan.n_pkts == 0
    empty = torch.empty(0, dtype=torch.uint8, device=txctx.device)
    assert plan.tx(empty).numel() == 0
    plan = txctx.plan_ragged(np.zeros(0, np.uint8), np.zeros(0, np.uint32))
    assert plan.n_samples == 0
    assert plan.modulate(plan.encode_bits(empty)).numel() == 0


def test_payload_stride_and_offsets(txctx, c_oracle):
    """Payloads need not be packed: a stride for uniform batches, arbitrary (also odd) byte offsets
    for ragged ones."""
    rng = np.random.default_rng(21)
    n, ln, stride = 6, 333, 401                       # odd stride: every other payload is misaligned
    buf = rng.integers(0, 256, n * stride, dtype=np.uint8)
    dev = txctx.device
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
    plan = txctx.plan_uniform(n, 4, ln, payload_stride=stride)
    iq = plan.tx(torch.from_numpy(buf).to(dev)).cpu().numpy()
    lay = plan.layout()
    for i in range(n):
        ref = c_oracle.encode_bits(buf[i * stride:i * stride + ln].tobytes(), 4)
        want = c_oracle.modulate(ref["interleaved"], 4, int(lay["pilot_base"][i]), 0.3, flags)["iq64"]
        a = int(lay["sample_off"][i])
        assert rel_l2_rows(iq[a:a + len(want)], want).max() <= REL_L2_TOL
    lens = np.array([10, 77, 1500, 3], np.uint32)
    offs = np.array([1, 131, 255, 2001], np.uint64)
    regs = np.array([1, 6, 8, 3], np.uint8)
    plan = txctx.plan_ragged(regs, lens, payload_off=offs)
    assert plan.payload_bytes == 2004
    iq = plan.tx(torch.from_numpy(buf).to(dev)).cpu().numpy()
    lay = plan.layout()
    for i in range(4):
        ref = c_oracle.encode_bits(buf[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes(), int(regs[i]))
        want = c_oracle.modulate(ref["interleaved"], int(regs[i]), int(lay["pilot_base"][i]), 0.3, flags)["iq64"]
        a = int(lay["sample_off"][i])
        assert rel_l2_rows(iq[a:a + len(want)], want).max() <= REL_L2_TOL


def test_side_stream(txctx):
    """Kernels go to the caller's stream: a side stream gives the same bits as the default one."""
    rng = np.random.default_rng(22)
    n, ln = 2000, 700
    payloads = torch.from_numpy(rng.integers(0, 256, n * ln, dtype=np.uint8)).to(txctx.device)
    plan = txctx.plan_uniform(n, 7, ln)
    ref = plan.tx(payloads)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=txctx.device)
    with torch.cuda.stream(side):
        got = plan.tx(payloads)
        got2 = plan.modulate(plan.encode_bits(payloads))
    side.synchronize()
    assert torch.equal(got.view(torch.float32), ref.view(torch.float32))
    assert torch.equal(got2.view(torch.float32), ref.view(torch.float32))


@pytest.mark.parametrize("role", ["A", "B"])
def test_pnc_roles(txctx, c_oracle, role):
    """PNC role plans (ftw/ftw_pnc_ofdm_preamble.cc:80-103, ftw/ftw_pnc_ofdm_pilot_cc.cc:125-178):
    six preamble rows, constant pilots; all rates, uniform and ragged, one chunk and several,
    fused and two-kernel paths, I/Q swap."""
    pf = _lib.F_PNC_A if role == "A" else _lib.F_PNC_B
    pm = _lib.PILOT_PNC_A if role == "A" else _lib.PILOT_PNC_B
    rng = np.random.default_rng(31)
    for regime, ln in [(1, 40), (2, 700), (3, 333), (4, 1), (5, 1500), (6, 0), (7, 2000), (8, 1500), (8, 3140)]:
        flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | pf | (_lib.F_SWAPIQ if regime == 6 else 0)
        payloads = rng.integers(0, 256, 3 * ln, dtype=np.uint8)
        plan = txctx.plan_uniform(3, regime, ln, pilot_mode=pm, flags=flags, amp=0.3)
        assert plan.n_samples == 3 * 80 * (6 + 1 + _lib.packet_info(regime, ln).n_sym + 13)
        check_plan(plan, payloads, [ln] * 3, [regime] * 3, c_oracle, flags | (pm << 8), 0.3)
    n = 60
    regs = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 1200, n).astype(np.uint32)
    payloads = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | pf
    plan = txctx.plan_ragged(regs, lens, pilot_mode=pm, flags=flags, amp=0.7)
    check_plan(plan, payloads, lens.tolist(), regs.tolist(), c_oracle, flags | (pm << 8), 0.7)


def test_pnc_mode_combinations(txctx, c_oracle):
    """Pilot block and preamble role are independent plan parameters: constant pilots with the
    standard preamble, no preamble at all, the all-ones pilot block (user 0); bad combinations."""
    rng = np.random.default_rng(32)
    ln = 500
    payloads = rng.integers(0, 256, 2 * ln, dtype=np.uint8)
    for pm, flags in [(_lib.PILOT_PNC_ALL, 3), (_lib.PILOT_PNC_A, 2), (_lib.PILOT_PNC_B, 0),
                      (_lib.PILOT_REF_STREAM, 1 | 2 | _lib.F_PNC_A), (_lib.PILOT_PER_PACKET, 1 | _lib.F_PNC_B)]:
        plan = txctx.plan_uniform(2, 5, ln, pilot_mode=pm, pilot_start=3, flags=flags, amp=0.3)
        check_plan(plan, payloads, [ln] * 2, [5] * 2, c_oracle, flags | ((pm if pm >= 2 else 0) << 8), 0.3)
    for pm, flags in [(5, 3), (-1, 3), (0, 3 | _lib.F_PNC_A | _lib.F_PNC_B), (0, 2 | _lib.F_PNC_A)]:
        with pytest.raises(_lib.OfdmTxError):
            txctx.plan_uniform(2, 5, ln, pilot_mode=pm, flags=flags)


def test_ragged_plan_reuse(txctx, c_oracle):
    """A ragged plan's waves draw packets from a counter the launcher resets: many more packets
    than waves, launched repeatedly (default stream, then a side stream), give the same samples
    every time, every packet exactly once; spot packets agree with the oracle."""
    rng = np.random.default_rng(23)
    n = 30000
    regs = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 400, n).astype(np.uint32)
    lens[::997] = 2304                                       # a few multi-chunk packets in between
    buf = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
    plan = txctx.plan_ragged(regs, lens, pilot_start=5, flags=flags, amp=0.3)
    pay = torch.from_numpy(buf).to(txctx.device)
    first = plan.tx(pay)
    for _ in range(3):
        assert torch.equal(plan.tx(pay).view(torch.float32), first.view(torch.float32))
    assert torch.equal(plan.modulate(plan.encode_bits(pay)).view(torch.float32), first.view(torch.float32))
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=txctx.device)
    with torch.cuda.stream(side):
        again = plan.tx(pay)
    side.synchronize()
    assert torch.equal(again.view(torch.float32), first.view(torch.float32))
    iq = first.cpu().numpy()
    lay = plan.layout()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for i in [0, 1, 996, 997, 998, n // 2, n - 2, n - 1]:
        ref = c_oracle.encode_bits(buf[offs[i]:offs[i + 1]].tobytes(), int(regs[i]))
        want = c_oracle.modulate(ref["interleaved"], int(regs[i]), int(lay["pilot_base"][i]), 0.3, flags)["iq64"]
        a = int(lay["sample_off"][i])
        assert rel_l2_rows(iq[a:a + len(want)], want).max() <= REL_L2_TOL


def test_ragged_short_packets(txctx, c_oracle):
    """Batches of very short packets: the waves then draw several packets per visit to the work
    counter (the last draw is a partial one).  Every packet against the oracle."""
    rng = np.random.default_rng(24)
    for n in (1, 7, 5003):
        regs = rng.integers(1, 9, n).astype(np.uint8)
        lens = rng.integers(0, 21, n).astype(np.uint32)
        buf = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
        flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP
        plan = txctx.plan_ragged(regs, lens, pilot_start=77, flags=flags, amp=0.3)
        check_plan(plan, buf, lens.tolist(), regs.tolist(), c_oracle, flags, 0.3)


@pytest.mark.parametrize("n", [0, 1, 2047, 2048, 2049, 70001])
def test_device_built_plan_equals_host_built_plan(txctx, n):
    """ofdmtx_plan_ragged_device (kernels: geometry, prefix sums, records) against the host walk
    of ofdmtx_plan_ragged: same sizes, same per-packet layout, same samples; with and without
    explicit payload offsets; block boundaries of the scan (2048 items) included."""
    rng = np.random.default_rng(25 + n)
    regs = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(0, 600, n).astype(np.uint32)
    if n > 2:
        lens[:3] = [0, 3140, 2304]
    for with_off in (False, True):
        off = None
        if with_off:
            gaps = rng.integers(0, 5, n).astype(np.uint64)
            off = (np.cumsum(lens.astype(np.uint64) + gaps) - lens).astype(np.uint64)      # increasing, with holes
        kw = dict(symbol_time=8.0, pilot_mode=_lib.PILOT_REF_STREAM, pilot_start=31, flags=3, amp=0.3)
        host = txctx.plan_ragged(regs, lens, payload_off=off, **kw)
        dev = txctx.plan_ragged(torch.from_numpy(regs).to(txctx.device), torch.from_numpy(lens.astype(np.int32)).to(txctx.device),
                                payload_off=None if off is None else torch.from_numpy(off.astype(np.int64)).to(txctx.device), **kw)
        for k in ("n_pkts", "payload_bytes", "plcp_bytes", "msg_bytes", "n_symbols", "n_rows", "n_samples"):
            assert getattr(host, k) == getattr(dev, k), k
        lh, ld = host.layout(), dev.layout()
        for k in lh:
            assert np.array_equal(lh[k], ld[k]), k
        if n:
            pay = torch.from_numpy(rng.integers(0, 256, max(host.payload_bytes, 1), dtype=np.uint8)).to(txctx.device)
            a, b = host.tx(pay), dev.tx(pay)
            assert torch.equal(a.view(torch.float32), b.view(torch.float32))
            # (tap buffers have unwritten padding between packets: compare the packets' bytes)
            ta = {k: v.cpu().numpy() for k, v in host.encode_bits(pay, taps=True).items()}
            tb = {k: v.cpu().numpy() for k, v in dev.encode_bits(pay, taps=True).items()}
            for i in list(range(min(n, 40))) + list(rng.integers(0, n, 40)):
                mo, ml, po, pn = (int(lh[k][i]) for k in ("msg_off", "msg_len", "plcp_off", "plcp_len"))
                assert all(np.array_equal(ta[k][mo:mo + ml], tb[k][mo:mo + ml]) for k in ("coded", "interleaved"))
                assert all(np.array_equal(ta[k][po:po + pn], tb[k][po:po + pn]) for k in ("plcp", "scrambled"))
    bad = torch.from_numpy(np.array([1, 9, 3], np.uint8)).to(txctx.device)
    with pytest.raises(_lib.OfdmTxError):
        txctx.plan_ragged(bad, torch.tensor([5, 5, 5], dtype=torch.int32, device=txctx.device))
    with pytest.raises(ValueError):
        txctx.plan_ragged(torch.from_numpy(np.array([1, 2], np.uint8)).to(txctx.device),
                          torch.tensor([5, 3141], dtype=torch.int32, device=txctx.device))


def test_custom_constellation(txctx, c_oracle):
    """The `constellation` argument of ftw.ofdm_mapper (ftw/ftw.i:157-163): replaced points show up
    in the mapper tap and in the IQ; the SIGNAL symbol keeps its fixed BPSK."""
    from ofdm_amd.batch import TxContext
    ctx = TxContext(0)                                 # own context: the shared one keeps the defaults
    rng = np.random.default_rng(23)
    pts = (rng.standard_normal(4) + 1j * rng.standard_normal(4)).astype(np.complex64)
    ctx.set_constellation(2, pts)
    n, ln = 3, 50
    payloads = rng.integers(0, 256, n * ln, dtype=np.uint8)
    plan = ctx.plan_uniform(n, 3, ln)
    dev = ctx.device
    msg = plan.encode_bits(torch.from_numpy(payloads).to(dev))
    iq, mapped = plan.modulate(msg, want_mapped=True)
    iq2 = plan.tx(torch.from_numpy(payloads).to(dev))
    assert torch.equal(iq.view(torch.float32), iq2.view(torch.float32))
    mapped = mapped.cpu().numpy()
    lay = plan.layout()
    for i in range(n):
        ref = c_oracle.encode_bits(payloads[i * ln:(i + 1) * ln].tobytes(), 3)
        bits = np.unpackbits(np.frombuffer(ref["interleaved"], np.uint8))
        so, ns = int(lay["sym_off"][i]), int(lay["n_sym"][i])
        assert np.array_equal(mapped[so], (2.0 * bits[:48] - 1).astype(np.complex64))
        idx = bits[48:48 + ns * 96].reshape(-1, 2) @ np.array([2, 1])
        assert np.array_equal(mapped[so + 1:so + 1 + ns].reshape(-1), pts[idx])
    with pytest.raises(_lib.OfdmTxError):
        ctx.set_constellation(3, pts)
    ctx.close()


@pytest.mark.parametrize("regime", range(1, 9))
def test_standard_receiver_decodes_gpu_frames(txctx, regime):
    """A receiver written from the 802.11a standard (tests/rx80211a.py) decodes what the fused
    kernel emits: SIGNAL parses to the right RATE/LENGTH, the PSDU carries the MAC header and the
    payload, the FCS checks -- at every rate, uniform and swapped I/Q."""
    import rx80211a as rx
    rng = np.random.default_rng(500 + regime)
    n, ln = 4, 83
    payloads = rng.integers(0, 256, n * ln, dtype=np.uint8)
    for flags, swap in ((_lib.F_PREAMBLE | _lib.F_ZEROGAP, False), (_lib.F_PREAMBLE | _lib.F_SWAPIQ, True)):
        plan = txctx.plan_uniform(n, regime, ln, pilot_start=77, flags=flags, amp=0.5)
        iq = plan.tx(torch.from_numpy(payloads).to(txctx.device)).cpu().numpy()
        per = plan.n_samples // n
        for i in range(n):
            got = rx.receive(iq[i * per:(i + 1) * per], gain=0.5, swap_iq=swap)
            assert got["length"] == 24 + ln + 4 and got["fcs_ok"] and not got["service"].any()
            assert got["psdu"][24:-4] == payloads[i * ln:(i + 1) * ln].tobytes()
            assert got["psdu"][:2] == b"\x08\x00"


@pytest.mark.parametrize("seed", range(12))
def test_randomised_plans(txctx, c_oracle, seed):
    """Random mixes of everything a plan can vary: rates, lengths (edge lengths over-represented),
    framing flags, pilot mode and start, amplitude, symbol time; ragged and uniform."""
    rng = np.random.default_rng(9000 + seed)
    flags = int(rng.integers(0, 8))
    amp = float(rng.choice([1.0, 0.3, 0.0625, 2.5]))
    pilot_mode = int(rng.integers(0, 2))
    pilot_start = int(rng.integers(0, 400))
    symbol_time = float(rng.choice([4.0, 8.0, 160.0]))
    n = int(rng.integers(1, 24))
    edge = np.array([0, 1, 2, 3, 4, 5, 23, 24, 25, 1499, 1500, 1501, 3139, 3140])
    lens = np.where(rng.random(n) < 0.4, rng.choice(edge, n), rng.integers(0, 600, n)).astype(np.uint32)
    regs = rng.integers(1, 9, n).astype(np.uint8)
    payloads = rng.integers(0, 256, int(lens.sum()) + 1, dtype=np.uint8)
    plan = txctx.plan_ragged(regs, lens, symbol_time=symbol_time, pilot_mode=pilot_mode, pilot_start=pilot_start,
                             flags=flags, amp=amp)
    check_plan(plan, payloads, [int(x) for x in lens], [int(r) for r in regs], c_oracle, flags, amp, symbol_time)
    ln, reg = int(lens[0]), int(regs[0])
    uni = rng.integers(0, 256, 3 * ln + 1, dtype=np.uint8)
    plan = txctx.plan_uniform(3, reg, ln, symbol_time=symbol_time, pilot_mode=pilot_mode, pilot_start=pilot_start,
                              flags=flags, amp=amp)
    check_plan(plan, uni, [ln] * 3, [reg] * 3, c_oracle, flags, amp, symbol_time)
