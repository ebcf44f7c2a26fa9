"""Block arguments and per-block taps of the FTW flowgraph on the GPU, through the C ABI:
preamble table / gap row passed to ftw.ofdm_preamble / ftw.zerogap, the `--log` files of
examples/ftw_ofdm.py:181-189 and examples/ftw_ofdm_tx_uhd.py:88-89, the guard on ragged plans and
the buffer checks of the batch front end.

Bar: pilot / cmap taps bit-exact; ifft / cp / scale / preamble taps and IQ within 1e-5 relative L2
per row against the float64 oracle.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from ofdm_amd import _lib
from oracle import ftw_oracle as fo

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-5


def rel_l2_rows(got, ref, n):
    got = np.asarray(got).reshape(-1, n).astype(np.complex128)
    ref = np.asarray(ref).reshape(-1, n)
    assert got.shape == ref.shape
    num = np.sqrt((np.abs(got - ref) ** 2).sum(axis=1))
    den = np.sqrt((np.abs(ref) ** 2).sum(axis=1))
    zero = den == 0
    assert np.all(num[zero] == 0)
    return np.where(zero, 0.0, num / np.where(zero, 1.0, den))


class Options(object):
    def __init__(self, **kw):
        self.regime, self.bandwidth, self.interp, self.swapIQ, self.log, self.role = "1", 20e6, 5.0, False, False, None
        self.tx_amplitude, self.to_file, self.from_file, self.tx_freq = 0.3, None, None, None
        self.__dict__.update(kw)


# ------------------------------------------------------------------ preamble table and gap row

@pytest.mark.parametrize("swap", [False, True])
def test_custom_preamble_table_and_gap_row(swap):
    """ftw_ofdm_preamble.cc:79-83 emits the table it is handed, ftw_zerogap.cc:88-92 the row it is
    handed: a random 4 x 80 table and a random non-zero gap row must come out, times the amplitude."""
    from ofdm_amd.batch import TxContext
    ctx = TxContext(0)                      # own context: the tables are per-context state
    rng = np.random.default_rng(21)
    table = (rng.standard_normal(330) + 1j * rng.standard_normal(330)).astype(np.complex64)   # rows beyond 4 x 80 unused
    gap = (rng.standard_normal(80) + 1j * rng.standard_normal(80)).astype(np.complex64)
    ctx.set_preamble(table)
    ctx.set_gap(gap)
    n, ln, regime, amp = 7, 211, 4, 0.37
    payloads = rng.integers(0, 256, n * ln, dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | (_lib.F_SWAPIQ if swap else 0)
    for plan in (ctx.plan_uniform(n, regime, ln, flags=flags, amp=amp),
                 ctx.plan_ragged(np.full(n, regime, np.uint8), np.full(n, ln, np.uint32), flags=flags, amp=amp)):
        iq = plan.tx(torch.from_numpy(payloads).to(ctx.device)).cpu().numpy()
        lay = plan.layout()
        for i in range(n):
            enc = fo.encode_bits(payloads[i * ln:(i + 1) * ln].tobytes(), str(regime))
            ref = fo.modulate(enc["interleaved"], str(regime), int(lay["pilot_base"][i]), amp, swap,
                              preamble_table=table, gap_row=gap)["iq"]
            a = int(lay["sample_off"][i])
            assert rel_l2_rows(iq[a:a + len(ref)], ref, 80).max() <= REL_L2_TOL
    # back to zeros: the fix-up kernel is skipped again and the gap rows are exact zeros
    ctx.set_gap(np.zeros(80, np.complex64))
    iq = ctx.plan_uniform(2, regime, ln).tx(torch.from_numpy(payloads).to(ctx.device)).cpu().numpy()
    assert np.all(iq.reshape(2, -1)[:, -13 * 80:] == 0)
    with pytest.raises(ValueError):
        ctx.set_preamble(table[:319])
    with pytest.raises(ValueError):
        ctx.set_gap(gap[:79])
    ctx.close()


def test_blocks_hand_their_tables_to_the_kernel(tmp_path):
    """The drop-in blocks pass what they hold: ofdm_mod with a preamble / zerogap block built from a
    different table produces that table (examples/ftw_packet_utils.py:374-392 builds the blocks)."""
    from ofdm_amd import ftw, ftw_ofdm, gr
    rng = np.random.default_rng(22)
    payload = rng.integers(0, 256, 60, dtype=np.uint8).tobytes()
    table = (rng.standard_normal(320) + 1j * rng.standard_normal(320)).astype(np.complex64)
    gap = (0.01 * rng.standard_normal(80) + 0j).astype(np.complex64)

    class Mod(ftw_ofdm.ofdm_mod):
        pass

    # the constructor reads the blocks from ftw_packet_utils.insert_preamble / insert_zerogap
    from ofdm_amd import ftw_packet_utils as pu
    orig_pre, orig_gap = pu.fft_preamble, pu.gap_sample
    pu.fft_preamble, pu.gap_sample = tuple(table.tolist()), tuple(gap.tolist())
    try:
        off0 = ftw_ofdm._pilot_offset[0]
        mod = Mod(Options(regime="3"), payload)
    finally:
        pu.fft_preamble, pu.gap_sample = orig_pre, orig_gap
    sink = gr.vector_sink_c()
    tb = gr.top_block()
    tb.connect(mod, gr.multiply_const_cc(0.5), sink)
    tb.start()
    mod.send_pkt(payload)
    mod.send_pkt(eof=True)
    tb.wait()
    enc = fo.encode_bits(payload, "3", mod._symbol_time)
    ref = fo.modulate(enc["interleaved"], "3", off0, 0.5, preamble_table=table, gap_row=gap)["iq"]
    assert rel_l2_rows(sink.data(), ref, 80).max() <= REL_L2_TOL
    with pytest.raises(ValueError):
        ftw.ofdm_preamble(80, 3, [list(table[:100])])
    with pytest.raises(RuntimeError):
        ftw.zerogap(64, 3, [list(gap)])


# ------------------------------------------------------------------ per-block taps

@pytest.mark.parametrize("regime,role", [(1, None), (4, None), (6, None), (8, None), (7, "A")])
def test_taps_behind_the_mapper(txctx, regime, role):
    rng = np.random.default_rng(30 + regime)
    lens = np.array([0, 37, 400, 1500, 5], dtype=np.uint32)
    n = len(lens)
    payloads = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    flags = _lib.F_PREAMBLE | _lib.F_ZEROGAP | ({None: 0, "A": _lib.F_PNC_A}[role])
    pilot_mode = _lib.PILOT_REF_STREAM if role is None else _lib.PILOT_PNC_A
    plan = txctx.plan_ragged(np.full(n, regime, np.uint8), lens, pilot_mode=pilot_mode, pilot_start=9, flags=flags,
                             amp=1.0)
    dev_pay = torch.from_numpy(payloads).to(txctx.device)
    msg = plan.encode_bits(dev_pay)
    iq, front = plan.modulate(msg, taps=True)
    back = plan.symbol_taps(iq)
    torch.cuda.synchronize()
    assert torch.equal(iq.view(torch.float32), plan.modulate(msg).view(torch.float32))      # the taps change nothing
    lay = plan.layout()
    front = {k: v.cpu().numpy() for k, v in front.items()}
    back = {k: v.cpu().numpy() for k, v in back.items()}
    off = 0
    pre_rows = 6 if role else 4
    for i in range(n):
        pl = payloads[off:off + int(lens[i])].tobytes()
        off += int(lens[i])
        enc = fo.encode_bits(pl, str(regime))
        ref = fo.modulate(enc["interleaved"], str(regime), int(lay["pilot_base"][i]), 1.0, role=role)
        so, ns = int(lay["sym_off"][i]), int(lay["n_sym"][i]) + 1
        for k in ("mapped", "pilot", "cmap"):                     # pure data movement: bit-exact
            want = ref[k].astype(np.complex64)
            assert np.array_equal(front[k][so:so + ns].view(np.uint64), want.view(np.uint64)), (k, i)
        for k, width in (("ifft", 64), ("cp", 80), ("scale", 80)):
            assert rel_l2_rows(back[k][so:so + ns], ref[k], width).max() <= REL_L2_TOL, (k, i)
        p0 = so + i * pre_rows
        want = ref["iq"][:(pre_rows + ns) * 80]
        assert rel_l2_rows(back["preamble"][p0:p0 + pre_rows + ns], want, 80).max() <= REL_L2_TOL
    # the taps need the path's own amplitude
    with pytest.raises(ValueError):
        txctx.plan_uniform(1, regime, 10, amp=0.3).symbol_taps(torch.zeros(80 * 40, dtype=torch.complex64, device=txctx.device))


def test_log_files_of_the_flowgraph(txctx, tmp_path, monkeypatch):
    """`--log`: every file of examples/ftw_ofdm.py:181-189 plus ftw_benchmark.dat
    (examples/ftw_ofdm_tx_uhd.py:88-89), with the reference's names and item sizes, byte-compatible
    with a gr.file_sink of those streams (raw little-endian complex64)."""
    from ofdm_amd import ftw_ofdm, ftw_ofdm_tx
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(6)
    payload = rng.integers(0, 256, 120, dtype=np.uint8).tobytes()
    opt = Options(regime="7", bandwidth=20e6, log=True, to_file=str(tmp_path / "out.dat"), swapIQ=True)
    off0 = ftw_ofdm._pilot_offset[0]
    tb = ftw_ofdm_tx.my_top_block(opt, payload)
    tb.start()
    n_pk = 3
    for _ in range(n_pk):
        tb.txpath.send_pkt(payload)
    tb.txpath.send_pkt(eof=True)
    tb.wait()
    refs, off = [], off0
    for _ in range(n_pk):
        r = fo.tx_packet(payload, "7", 4.0, pilot_offset=off, amp=1.0)
        off += r["info"]["N_sym"] + 1
        refs.append(r)
    ns = refs[0]["info"]["N_sym"] + 1
    cat = lambda k: np.concatenate([np.asarray(r[k]).reshape(-1) for r in refs])
    rd = lambda name: np.fromfile(name, np.complex64)
    assert np.array_equal(rd("ofdm_mapper.dat").view(np.uint64), cat("mapped").astype(np.complex64).view(np.uint64))
    assert np.array_equal(rd("ofdm_pilot.dat").view(np.uint64), cat("pilot").astype(np.complex64).view(np.uint64))
    assert np.array_equal(rd("ofdm_cmap.dat").view(np.uint64), cat("cmap").astype(np.complex64).view(np.uint64))
    assert rd("ofdm_pilot.dat").size == n_pk * ns * 53 and rd("ofdm_cmap.dat").size == n_pk * ns * 64
    assert rel_l2_rows(rd("ofdm_ifft.dat"), cat("ifft"), 64).max() <= REL_L2_TOL
    assert rel_l2_rows(rd("ofdm_cp_adder.dat"), cat("cp"), 80).max() <= REL_L2_TOL
    assert rel_l2_rows(rd("ofdm_scale.dat"), cat("scale"), 80).max() <= REL_L2_TOL
    pre = np.concatenate([r["iq"][:(4 + ns) * 80] for r in refs])
    assert rel_l2_rows(rd("ofdm_preamble.dat"), pre, 80).max() <= REL_L2_TOL
    assert rel_l2_rows(rd("ofdm_zerogap.dat"), cat("iq"), 80).max() <= REL_L2_TOL
    # the path's output in front of tx_amplitude, I and Q exchanged as the path was asked to
    bench = rd("ftw_benchmark.dat")
    unit = cat("iq")
    assert rel_l2_rows(bench, unit.imag + 1j * unit.real, 80).max() <= REL_L2_TOL
    out = rd(str(tmp_path / "out.dat"))
    assert rel_l2_rows(out, float(np.float32(0.3)) * (unit.imag + 1j * unit.real), 80).max() <= REL_L2_TOL


# ------------------------------------------------------------------ guard rails

def test_ragged_plan_on_a_second_stream_while_busy(txctx):
    """The work counters of a ragged plan are shared by its launches: a launch on another stream while the
    previous one still runs is ordered behind it on the device (not refused, not raced) -- both launches
    produce the batch."""
    rng = np.random.default_rng(40)
    n = 200000
    regimes = rng.integers(1, 9, n).astype(np.uint8)
    lens = rng.integers(200, 1500, n).astype(np.uint32)
    plan = txctx.plan_ragged(regimes, lens)
    payloads = torch.randint(0, 256, (int(lens.sum()),), dtype=torch.uint8, device=txctx.device)
    want = plan.tx(payloads)                                    # (the first launch of a kernel in a process loads its code)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=txctx.device)
    a, b, c = plan.alloc_iq(), plan.alloc_iq(), plan.alloc_iq()
    for t in (a, b, c):
        t.zero_()
    torch.cuda.synchronize()
    plan.tx(payloads, out=a)                                    # two milliseconds of work
    with torch.cuda.stream(side):
        plan.tx(payloads, out=b)                                # another stream, the first launch still running
        msg = plan.encode_bits(payloads)                        # the bit chain draws from counters of the same plan
    plan.tx(payloads, out=c)                                    # and back on the first stream
    torch.cuda.synchronize()
    for t in (a, b, c):
        assert torch.equal(t.view(torch.float32), want.view(torch.float32))
    assert torch.equal(plan.modulate(msg).view(torch.float32), want.view(torch.float32))
    del a, b, c, msg
    # uniform plans have no shared state
    uni = txctx.plan_uniform(20000, 8, 1500)
    pay = torch.zeros(uni.payload_bytes, dtype=torch.uint8, device=txctx.device)
    out = uni.alloc_iq()
    uni.tx(pay, out=out)
    with torch.cuda.stream(side):
        uni.tx(pay, out=out)
    torch.cuda.synchronize()


def test_undersized_or_foreign_buffers_are_refused(txctx):
    plan = txctx.plan_uniform(4, 8, 100)
    pay = torch.zeros(plan.payload_bytes, dtype=torch.uint8, device=txctx.device)
    good = plan.alloc_iq()
    for bad in (good[:-1], good.view(torch.float32)[:plan.n_samples], good.cpu(), good[::2],
                torch.zeros(plan.n_samples, dtype=torch.complex128, device=txctx.device)):
        with pytest.raises(ValueError):
            plan.tx(pay, out=bad)
        with pytest.raises(ValueError):
            plan.modulate(plan.alloc_msg(), out=bad)
    with pytest.raises(ValueError):
        plan.tx(pay[:-1])
    with pytest.raises(ValueError):
        plan.rx_demod(good[:-1])
    with pytest.raises(ValueError):
        plan.rx_demod(good, out=plan.alloc_msg()[:-1])
    with pytest.raises(ValueError):
        plan.rx_decode(plan.alloc_msg(), status=torch.zeros(3, dtype=torch.int32, device=txctx.device))
    with pytest.raises(ValueError):
        plan.encode_bits(pay, out=plan.alloc_msg()[:-1])


def test_two_modulators_keep_their_own_constellations():
    """The mapper's constellation is an argument of the block instance (ftw/ftw.i:157-163): two
    ofdm_mod objects with different point sets of one order must not overwrite each other."""
    from ofdm_amd import ftw_ofdm, gr
    from ofdm_amd import ftw_packet_utils as pu
    payload = bytes(range(50))
    a = ftw_ofdm.ofdm_mod(Options(regime="3"), payload)
    rot = tuple(complex(c) * 1j for c in pu.gray_const_qpsk)
    orig = pu.qpsk
    pu.qpsk = lambda self: rot
    try:
        b = ftw_ofdm.ofdm_mod(Options(regime="3"), payload)
    finally:
        pu.qpsk = orig
    outs = []
    for m in (a, b, a):
        sink = gr.vector_sink_c()
        m._outs = [(None, sink)]
        ftw_ofdm._pilot_offset[0] = 0
        m.send_pkt(payload)
        m.send_pkt(eof=True)
        outs.append(sink.data())
    assert np.array_equal(outs[0], outs[2])
    assert not np.array_equal(outs[0], outs[1])
    enc = fo.encode_bits(payload, "3", a._symbol_time)
    assert rel_l2_rows(outs[0], fo.modulate(enc["interleaved"], "3", 0, 1.0)["iq"], 80).max() <= REL_L2_TOL


def test_probed_output_placement_keeps_one_buffer(txctx):
    """TxPlan.alloc_iq_probed: the caller's launch runs into every candidate, one buffer of the plan's
    size comes back with the candidates' times, the others return to the device, and the kept buffer
    holds a complete result."""
    plan = txctx.plan_uniform(4000, 8, 1500)
    pay = torch.randint(0, 256, (plan.payload_bytes,), dtype=torch.uint8, device=txctx.device)
    want = plan.tx(pay)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(txctx.device)
    calls = []

    def fill(b):
        calls.append(b.data_ptr())
        plan.tx(pay, out=b)

    iq, ms = plan.alloc_iq_probed(fill, max_candidates=3, reserve_bytes=1 << 30, launches=1)
    torch.cuda.synchronize()
    assert len(ms) == 3 and all(t > 0 for t in ms) and len(set(calls)) == 3 and len(calls) == 6
    assert iq.numel() == plan.n_samples and iq.data_ptr() in calls
    assert torch.equal(iq.view(torch.float32), want.view(torch.float32))
    free1, _ = torch.cuda.mem_get_info(txctx.device)
    assert free0 - free1 <= plan.n_samples * 8 + (64 << 20)          # one buffer kept (allocator rounding aside)


def test_spread_output_buffer(txctx):
    """ofdmtx_stream_alloc / ofdmtx_stream_free through TxContext.alloc_spread: the buffer is usable as
    d_iq like any device memory (same samples as into an ordinary allocation, device-to-host copy
    included), its memory returns to the device when the last tensor viewing it is gone, and a pointer
    the library did not hand out is refused."""
    plan = txctx.plan_uniform(3000, 8, 1500)
    pay = torch.randint(0, 256, (plan.payload_bytes,), dtype=torch.uint8, device=txctx.device)
    want = plan.tx(pay)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info(txctx.device)
    iq, spread = plan.alloc_iq_spread()
    assert iq.dtype == torch.complex64 and iq.numel() == plan.n_samples and iq.device == txctx.device
    if free0 > 0.8 * total:
        assert spread, "a nearly empty device must provide the layout"
        # 144 MB = 3 pieces of 64 MiB; memory was taken 16 GiB (256 handles) at a time until a second part showed up
        cost = txctx.last_alloc
        assert cost["spread"] and cost["parts"] in (2, 3) and cost["handles"] % 256 == 0
        assert (16 << 30) <= cost["peak_held_bytes"] == cost["handles"] * (64 << 20) < free0
    iq.zero_()
    plan.tx(pay, out=iq)
    torch.cuda.synchronize()
    assert torch.equal(iq.view(torch.float32), want.view(torch.float32))
    assert np.array_equal(iq[:4096].cpu().numpy(), want[:4096].cpu().numpy())
    part = iq[100:200]                                               # a view keeps the memory alive
    del iq
    assert torch.equal(part, want[100:200])
    del part
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(txctx.device)
    # (the driver keeps some tens of MB of page tables from the search; one lost piece would be 64 MiB more)
    assert free0 - free1 < (64 << 20), "the buffer's memory (3 x 64 MiB) did not return to the device"
    lib = txctx._lib
    assert lib.ofdmtx_stream_free(txctx._h, C.c_void_p(want.data_ptr())) == _lib.EINVAL
    ptr = C.c_void_p()
    assert lib.ofdmtx_stream_alloc(txctx._h, 0, C.byref(ptr), None) == _lib.EINVAL


def test_spread_allocation_falls_back_when_the_device_is_occupied(txctx):
    """With most of the device taken no second part shows up within the free memory: the call still
    returns usable memory (an ordinary allocation) and says so; two buffers can be live at once."""
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info(txctx.device)
    if free0 < 0.8 * total:
        pytest.skip("the device is not empty enough to stage this")
    first, spread1 = txctx.alloc_spread(1 << 30)
    assert spread1 and txctx.last_alloc["parts"] >= 2
    free_now, _ = torch.cuda.mem_get_info(txctx.device)
    hog = torch.empty(free_now - (6 << 30), dtype=torch.uint8, device=txctx.device)    # all but 6 GiB: too little to look for parts in
    second, spread2 = txctx.alloc_spread(1 << 30)
    assert not spread2 and txctx.last_alloc["parts"] == 0
    first.fill_(7)
    second.fill_(9)
    torch.cuda.synchronize()
    assert int(first[-1]) == 7 and int(second[0]) == 9 and int(second[-1]) == 9
    del first, second, hog
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(txctx.device)
    assert free0 - free1 < (64 << 20)                  # (two buffers of 1 GiB were live)


def test_spread_allocation_with_three_parts(txctx):
    """ofdmtx_stream_alloc_parts(want_parts = 3) on an empty device: pieces of all three parts, found further
    into the memory than two; the buffer works like any other."""
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info(txctx.device)
    if free0 < 0.9 * total:
        pytest.skip("the device is not empty enough to reach the third part")
    buf, spread = txctx.alloc_spread(3 << 30, parts=3)
    cost = txctx.last_alloc
    # (a device with a history may show all three parts early on; one that has only just been handed over
    #  shows them in turn, the third about 192 GiB in; two parts only after the whole free memory was looked at)
    assert spread and (cost["parts"] == 3 or (cost["parts"] == 2 and cost["peak_held_bytes"] > 0.9 * free0 - (4 << 30)))
    buf.fill_(5)
    torch.cuda.synchronize()
    assert int(buf[0]) == 5 and int(buf[-1]) == 5 and buf.numel() == 3 << 30
    ptr = C.c_void_p()
    assert txctx._lib.ofdmtx_stream_alloc_parts(txctx._h, 1 << 20, 4, C.byref(ptr), None) == _lib.EINVAL


def test_spread_allocation_on_a_device_with_a_history(txctx):
    """Memory that was taken whole and given back in a random order comes out of the driver in no order:
    windows of the allocation order are then mixtures of parts or lie in any of them.  The parts are still
    found (they are measured, not assumed), and the buffer takes stores at the rate of a spread one."""
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info(txctx.device)
    if free0 < 0.9 * total:
        pytest.skip("needs the device to itself")
    rng = np.random.default_rng(91)
    n = int((free0 - (40 << 30)) >> 30)
    held = [torch.empty(1 << 30, dtype=torch.uint8, device=txctx.device) for _ in range(n)]
    for i in rng.permutation(n):
        held[i] = None
        if i % 16 == 0:
            torch.cuda.empty_cache()                    # back to the driver a few at a time, in that order
    del held
    torch.cuda.empty_cache()
    buf, spread = txctx.alloc_spread(4 << 30)
    assert spread and txctx.last_alloc["parts"] >= 2
    buf.fill_(3)
    torch.cuda.synchronize()
    assert int(buf[0]) == 3 and int(buf[-1]) == 3
    del buf


def test_output_and_input_in_one_spread_buffer(txctx):
    """TxPlan.alloc_iq_spread(input_bytes=...): the payloads live behind the output in the same buffer; the
    launch gives the samples it gives from an ordinary allocation."""
    n, ln = 300, 700
    plan = txctx.plan_uniform(n, 6, ln)
    pay = torch.randint(0, 256, (n * ln,), dtype=torch.uint8, device=txctx.device)
    want = plan.tx(pay)
    iq, room, spread = plan.alloc_iq_spread(input_bytes=n * ln)
    assert iq.numel() == plan.n_samples and room.numel() == n * ln and room.data_ptr() % (64 << 20) == iq.data_ptr() % (64 << 20)
    room.copy_(pay)
    got = plan.tx(room, out=iq)
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    plan.close()
