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
        enc = c_oracle.encode_bits(pl, int(r), symbol_time)
        mod = c_oracle.modulate(enc["interleaved"], int(r), int(pilot_base[i]), amp, flags)
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


def check_plan(plan, payloads_np, lens, regimes, c_oracle, flags, amp):
    dev = plan.ctx.device
    payloads = torch.from_numpy(payloads_np).to(dev)
    taps = plan.encode_bits(payloads, taps=True)
    iq, mapped = plan.modulate(taps["interleaved"], want_mapped=True)
    iq2 = plan.tx(payloads)
    torch.cuda.synchronize()
    lay = plan.layout()
    ref = oracle_batch(c_oracle, payloads_np, lens, regimes, lay["pilot_base"], flags, amp)
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
