"""The Python drop-in layer (ofdm_amd.ftw_packet_utils / ftw / gr / ftw_ofdm / ftw_ofdm_tx) used
the way the reference's scripts use their originals (examples/ftw_ofdm.py:191-219,
examples/ftw_ofdm_tx_uhd.py:96-175), checked against the oracle.

Bar: bytes bit-exact; IQ within 1e-5 relative L2 per 80-sample row.
"""
import os
import threading

import numpy as np
import pytest

from oracle import ftw_oracle as fo

REL_L2_TOL = 1e-5


class Options(object):
    """What examples/ftw_ofdm_tx_uhd.py:101-141 parses, with its defaults."""

    def __init__(self, **kw):
        self.regime = "1"
        self.bandwidth = 500e3
        self.interp = None
        self.swapIQ = False
        self.log = False
        self.role = None
        self.tx_amplitude = 0.3
        self.to_file = None
        self.from_file = None
        self.tx_freq = None
        self.__dict__.update(kw)


# ------------------------------------------------------------------------------ CPU

@pytest.mark.parametrize("regime", [str(r) for r in range(1, 9)])
def test_get_info_matches_oracle(regime):
    from ofdm_amd import ftw_packet_utils as pu
    for n, t in ((0, 4.0), (1, 4.0), (100, 8.0), (1500, 4.0), (3140, 160.0), (3141, 4.0), (5000, 4.0)):
        payload = bytes(range(256)) * 20
        a, b = pu.get_info(payload[:n], regime, t), fo.get_info(payload[:n], regime, t)
        assert set(a) == set(b)
        for k in b:
            assert a[k] == b[k], (k, n)


def test_get_info_unknown_regime():
    from ofdm_amd import ftw_packet_utils as pu
    with pytest.raises(UnboundLocalError):             # examples/ftw_packet_utils.py:39-78 has no else
        pu.get_info(b"abc", "9", 4.0)


def test_bit_string_helpers():
    from ofdm_amd import ftw_packet_utils as pu
    assert pu.conv_packed_binary_string_to_1_0_string(b"\xAF") == "10101111"
    assert pu.conv_1_0_string_to_packed_binary_string("10101111") == b"\xAF"
    assert pu.conv_1_0_string_to_packed_binary_string("111") == b"\x07"      # left-padded
    with pytest.raises(ValueError):
        pu.conv_1_0_string_to_packed_binary_string("10201")
    assert pu.is_1_0_string("0101") and not pu.is_1_0_string(b"0101") and not pu.is_1_0_string("012")
    assert pu.hexint(0xf0f0f0f2) == 0xf0f0f0f2 - 2 ** 32 and pu.hexint(5) == 5
    assert pu.ascii_to_bin("A") == "01000001"
    assert pu.string_to_hex_list(b"\x01\xff") == ["0x1", "0xff"]
    assert pu.make_MPDU(b"\x01\x80\xf0") == b"\x80\x01\x0f"
    pkt = bytes(range(40))
    want = fo._bitrev8(pkt) + fo.ftw_crc32(pkt).to_bytes(4, "big")
    assert pu.gen_and_append_crc32(pu.make_MPDU(pkt), pkt) == want


def test_static_tables_match_reference_literals():
    import json
    from ofdm_amd import ftw_packet_utils as pu
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ftw_tables.json")))
    for name in ("gray_const_bpsk", "gray_const_qpsk", "gray_const_qam16", "gray_const_qam64", "fft_preamble"):
        want = np.array([complex(re, im) for re, im in gold[name]]).astype(np.complex64)
        got = np.array(getattr(pu, name), dtype=np.complex64)
        assert np.array_equal(got, want), name
    assert pu.gap_sample == (0j,) * 80


def test_msg_queue_blocks_at_limit():
    from ofdm_amd import gr
    q = gr.msg_queue(2)
    q.insert_tail(gr.message_from_string(b"a"))
    q.insert_tail(gr.message_from_string(b"b"))
    assert q.full_p()
    done = []
    t = threading.Thread(target=lambda: (q.insert_tail(gr.message(1)), done.append(1)))
    t.start()
    t.join(0.2)
    assert not done                                   # producer waits, as send_pkt does at limit 2
    assert q.delete_head().to_string() == b"a"
    t.join(2)
    assert done and q.count() == 2
    assert q.delete_head().to_string() == b"b" and q.delete_head().type() == 1 and q.empty_p()


def test_block_parameter_checks():
    from ofdm_amd import ftw
    with pytest.raises(RuntimeError):                  # ftw_ofdm_mapper.cc:61-62
        ftw.ofdm_mapper((-1, 1), 2, 65, 64)
    with pytest.raises(RuntimeError):
        ftw.ofdm_mapper((-1, 1), 2, 52, 64)
    with pytest.raises(RuntimeError):                  # ftw_ofdm_pilot_cc.cc:110-111
        ftw.ofdm_pilot_cc(52).check()
    with pytest.raises(RuntimeError):                  # ftw_ofdm_cmap_cc.cc:97-99
        ftw.ofdm_cmap_cc(128, 53).check()
    m = ftw.ofdm_mapper((-1, 1), 2, 48, 64)
    assert m.msgq().limit() == 2 and m.nbits == 1
    assert ftw.ftw_crc32("123456789") == fo.ftw_crc32(b"123456789")


# ------------------------------------------------------------------------------ GPU

def rel_l2_rows(got, ref):
    got = np.asarray(got).reshape(-1, 80).astype(np.complex128)
    ref = np.asarray(ref).reshape(-1, 80)
    assert got.shape == ref.shape
    num = np.sqrt((np.abs(got - ref) ** 2).sum(axis=1))
    den = np.sqrt((np.abs(ref) ** 2).sum(axis=1))
    assert np.all(num[den == 0] == 0)
    return np.where(den == 0, 0.0, num / np.where(den == 0, 1.0, den))


@pytest.mark.gpu
@pytest.mark.parametrize("regime", [str(r) for r in range(1, 9)])
def test_send_pkt_call_sequence(txctx, regime):
    """The body of ofdm_mod.send_pkt (examples/ftw_ofdm.py:201-214), stage by stage."""
    from ofdm_amd import ftw_packet_utils as pu
    rng = np.random.default_rng(int(regime))
    for n in (0, 1, 10, 257, 1500):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        info = pu.get_info(payload, regime, 4.0)
        N_cbps, N_bpsc, N_rate, N_sym = info["N_cbps"], info["N_bpsc"], info["rate"], info["N_sym"]
        (pkt, Length) = pu.ftw_make(payload, regime, 4.0)
        (pkt_scrambled, Length) = pu.scrambler(pkt, Length)
        pkt_coded = pu.conv_encoder(pkt_scrambled, Length, regime, N_cbps, N_bpsc, N_sym, N_rate)
        pkt_interleaved = pu.interleaver(pkt_coded, regime, N_cbps, N_bpsc)
        ref = fo.encode_bits(payload, regime, 4.0)
        assert Length == ref["Length"]
        assert pkt == ref["plcp"], "ftw_make"
        assert pkt_scrambled == ref["scrambled"], "scrambler"
        assert pkt_coded == ref["coded"], "conv_encoder"
        assert pkt_interleaved == ref["interleaved"], "interleaver"


@pytest.mark.gpu
def test_stage_operators_on_arbitrary_strings(txctx):
    """The stage functions accept any byte string of a valid shape, not only ftw_make's output."""
    from ofdm_amd import ftw_packet_utils as pu
    rng = np.random.default_rng(11)
    for regime in ("2", "5", "7", "8"):
        n_cbps, n_bpsc = {"2": (48, 1), "5": (192, 4), "7": (288, 6), "8": (288, 6)}[regime]
        n_sym = 7
        n_dbps = int(n_cbps * {"2": 0.75, "5": 0.5, "7": 2 / 3.0, "8": 0.75}[regime])
        raw = rng.integers(0, 256, 3 + (n_sym * n_dbps + 7) // 8 + 2, dtype=np.uint8).tobytes()
        length = len(raw) - 5 - 1
        assert pu.scrambler(raw, length) == fo.scrambler(raw, length)
        assert pu.conv_encoder(raw, length, regime, n_cbps, n_bpsc, n_sym, None) == \
            fo.conv_encoder(raw, length, regime, n_cbps, n_bpsc, n_sym, None)
        coded = rng.integers(0, 256, 6 + n_sym * n_cbps // 8, dtype=np.uint8).tobytes()
        assert pu.interleaver(coded, regime, n_cbps, n_bpsc) == fo.interleaver(coded, regime, n_cbps, n_bpsc)
    with pytest.raises(IndexError):
        pu.scrambler(b"\x00" * 10, 10)
    with pytest.raises(ValueError):
        pu.conv_encoder(b"\x00" * 10, 5, "8", 288, 6, 50, 0.75)
    with pytest.raises(ValueError):
        pu.interleaver(b"\x00" * 11, "8", 288, 6)
    with pytest.raises(ValueError):
        pu.ftw_make(b"\x00" * 3141, "1", 4.0)          # examples/ftw_packet_utils.py:119-121


@pytest.mark.gpu
def test_c1_one_packet_6mbps_to_file(txctx, tmp_path):
    """One 1500-byte packet at 6 Mbit/s into a file sink, 20 MHz: the reference's own runnable case."""
    from ofdm_amd import ftw_ofdm, ftw_ofdm_tx
    rng = np.random.default_rng(1)
    payload = rng.integers(0, 256, 1500, dtype=np.uint8).tobytes()
    out = str(tmp_path / "c1.dat")
    opt = Options(regime="1", bandwidth=20e6, to_file=out)
    off0 = ftw_ofdm._pilot_offset[0]
    tb = ftw_ofdm_tx.my_top_block(opt, payload)
    assert opt.interp == 5.0 and tb.txpath.ofdm_tx._symbol_time == 4.0
    tb.start()
    tb.txpath.send_pkt(payload, eof=False)
    tb.txpath.send_pkt(eof=True)
    tb.wait()
    got = np.fromfile(out, dtype=np.complex64)
    assert got.size == 42320
    ref = fo.tx_packet(payload, "1", 4.0, pilot_offset=off0, amp=0.3)["iq"]
    assert rel_l2_rows(got, ref).max() <= REL_L2_TOL
    assert ftw_ofdm._pilot_offset[0] == off0 + 512


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [False, True], ids=["allocations", "ring"])
@pytest.mark.parametrize("regime,swap", [("8", False), ("6", True), ("3", False)])
def test_stream_of_packets_keeps_pilot_counter(txctx, tmp_path, monkeypatch, regime, swap, ring):
    """ring: every batch counts as large -- it is written into a slot of the output ring (a buffer whose
    pieces alternate between the parts of the device memory) and handed to the sinks one batch late."""
    from ofdm_amd import batch, ftw_ofdm, ftw_ofdm_tx, gr
    if ring:
        monkeypatch.setattr(batch.StreamOutput, "RING_MIN_BYTES", 1)
    rng = np.random.default_rng(3)
    lens = [1500, 1500, 40, 733, 0, 1500]
    payloads = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    opt = Options(regime=regime, bandwidth=20e6, swapIQ=swap, tx_amplitude=0.25)
    off0 = ftw_ofdm._pilot_offset[0]
    tb = ftw_ofdm_tx.my_top_block(opt, payloads[0])
    sink = gr.vector_sink_c()
    tb.connect(tb.txpath, tb.amp, sink)
    tb.txpath.ofdm_tx._batch_packets = 4 if not ring else 2     # two batches: 4 + 2 packets (ring: three of 2)
    tb.start()
    for i, p in enumerate(payloads):
        tb.txpath.send_pkt(p)
        if ring and i == 3:                            # two batches enqueued, the first handed over
            assert tb.sink.items_seen == sum(ftw_ofdm.ofdm_packet_utils.get_info(q, regime, 4.0)["N_sym"] + 18
                                             for q in payloads[:2]) * 80
    tb.txpath.send_pkt(eof=True)
    tb.wait()
    out = tb.txpath.ofdm_tx._output
    assert (out.ring is not None) == ring and (not ring or (out.ring.slots == 2 and out.ring.slot_bytes == 192 << 20))
    off = off0
    ref = []
    for p in payloads:
        r = fo.tx_packet(p, regime, 4.0, pilot_offset=off, amp=0.25, swap_iq=swap)
        off += r["info"]["N_sym"] + 1
        ref.append(r["iq"])
    ref = np.concatenate(ref)
    got = sink.data()
    assert got.size == ref.size
    assert rel_l2_rows(got, ref).max() <= REL_L2_TOL
    assert tb.sink.items_seen == ref.size


@pytest.mark.gpu
def test_log_taps(txctx, tmp_path, monkeypatch):
    from ofdm_amd import ftw_ofdm, ftw_ofdm_tx
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    payload = rng.integers(0, 256, 300, dtype=np.uint8).tobytes()
    opt = Options(regime="5", bandwidth=10e6, log=True)            # 802.11p: symbol time 8 us
    off0 = ftw_ofdm._pilot_offset[0]
    tb = ftw_ofdm_tx.my_top_block(opt, payload)
    tb.start()
    tb.txpath.send_pkt(payload)
    tb.txpath.send_pkt(eof=True)
    tb.wait()
    ref = fo.tx_packet(payload, "5", 8.0, pilot_offset=off0, amp=1.0)
    n = len(ref["interleaved"])
    assert np.fromfile("ofdm_interleaved.dat", np.uint8)[:n].tobytes() == ref["interleaved"]
    assert np.fromfile("ofdm_coded.dat", np.uint8)[:n].tobytes() == ref["coded"]
    assert np.fromfile("ofdm_scrambled.dat", np.uint8)[:len(ref["plcp"])].tobytes() == ref["scrambled"]
    mapped = np.fromfile("ofdm_mapper.dat", np.complex64).reshape(-1, 48)
    assert np.array_equal(mapped, ref["mapped"].astype(np.complex64))
    unit = np.fromfile("ofdm_zerogap.dat", np.complex64)
    assert rel_l2_rows(unit, ref["iq"]).max() <= REL_L2_TOL


@pytest.mark.gpu
def test_send_pkt_rejects_long_payload(txctx):
    from ofdm_amd import ftw_ofdm
    mod = ftw_ofdm.ofdm_mod(Options(regime="1", interp=5.0), b"")
    with pytest.raises(ValueError):
        mod.send_pkt(b"\x00" * 3141)


@pytest.mark.gpu
@pytest.mark.parametrize("role,regime", [("A", "1"), ("B", "1"), ("A", "6"), ("B", "8")])
def test_pnc_roles_cli(txctx, tmp_path, role, regime):
    """--role A / B (examples/ftw_ofdm_tx_uhd.py:143-148, examples/ftw_ofdm.py:101-106,133-149):
    the PNC pilot, preamble and zero-gap blocks; role B sends "WorldHello"."""
    from ofdm_amd import ftw_ofdm_tx
    out = str(tmp_path / "pnc.dat")
    ftw_ofdm_tx.main(["--to-file", out, "--regime", regime, "-W", "20e6", "-N", "3", "-r", "11", "--role", role])
    payload = (b"HelloWorld" if role == "A" else b"WorldHello") * 11
    one = fo.tx_packet(payload, regime, 4.0, amp=0.3, role=role)
    assert one["iq"].size == 80 * (6 + 1 + one["info"]["N_sym"] + 13)
    got = np.fromfile(out, dtype=np.complex64)
    # constant pilots: every packet of the stream is the same
    assert rel_l2_rows(got, np.tile(one["iq"], 3)).max() <= REL_L2_TOL


@pytest.mark.gpu
def test_pnc_blocks_and_repetition(txctx):
    from ofdm_amd import _lib, ftw, ftw_packet_utils as pu
    import torch
    assert isinstance(pu.insert_preamble(80, 10, 'A'), ftw.pnc_ofdm_preamble)
    assert pu.insert_preamble(80, 10, 'B').flags == _lib.F_PREAMBLE | _lib.F_PNC_B
    assert type(pu.insert_preamble(80, 10)) is ftw.ofdm_preamble
    assert isinstance(pu.insert_zerogap(80, 10, 'FPNC'), ftw.pnc_zerogap)
    assert type(pu.insert_zerogap(80, 10)) is ftw.zerogap
    assert [ftw.pnc_ofdm_pilot_cc(48, u).pilot_mode for u in (1, 2, 0)] == [_lib.PILOT_PNC_A, _lib.PILOT_PNC_B,
                                                                           _lib.PILOT_PNC_ALL]
    with pytest.raises(RuntimeError):
        ftw.pnc_ofdm_pilot_cc(52, 1).check()
    with pytest.raises(RuntimeError):
        ftw.pnc_ofdm_preamble(80, 10, [list(pu.fft_preamble)], 3)
    # ftw.repetition: the first (N_sym + 18) rows of the input, again and again
    n_sym = 7
    rep = ftw.repetition(80, 0, n_sym)
    assert rep.period == 80 * 25
    rng = np.random.default_rng(9)
    src = (rng.standard_normal(80 * 60) + 1j * rng.standard_normal(80 * 60)).astype(np.complex64)
    dev = torch.from_numpy(src).to(txctx.device)
    for rows in (1, 25, 26, 100, 1003):
        got = rep.work(dev, rows).cpu().numpy()
        want = np.tile(src[:rep.period], rows // 25 + 1)[:80 * rows]
        assert np.array_equal(got, want), rows
    with pytest.raises(ValueError):
        rep.work(dev[:100], 5)


@pytest.mark.gpu
def test_cli_main(txctx, tmp_path):
    from ofdm_amd import ftw_ofdm, ftw_ofdm_tx
    out = str(tmp_path / "cli.dat")
    off0 = ftw_ofdm._pilot_offset[0]
    ftw_ofdm_tx.main(["--to-file", out, "--regime", "4", "-W", "20e6", "-N", "3", "-r", "7"])
    payload = b"HelloWorld" * 7
    ref, off = [], off0
    for _ in range(3):
        r = fo.tx_packet(payload, "4", 4.0, pilot_offset=off, amp=0.3)
        off += r["info"]["N_sym"] + 1
        ref.append(r["iq"])
    got = np.fromfile(out, dtype=np.complex64)
    assert rel_l2_rows(got, np.concatenate(ref)).max() <= REL_L2_TOL


@pytest.mark.gpu
def test_replay_from_file(txctx, tmp_path, monkeypatch):
    """--log leaves the unit-amplitude stream in ofdm_zerogap.dat (the file the reference's burst
    replayer reads, my_benchmark_ofdm/tx_file_bursts.cpp:56); --from-file sends it through
    tx_amplitude again (examples/ftw_ofdm_tx_uhd.py:75-86): same samples as a direct run."""
    from ofdm_amd import ftw_ofdm, ftw_ofdm_tx
    monkeypatch.chdir(tmp_path)
    off0 = ftw_ofdm._pilot_offset[0]
    ftw_ofdm_tx.main(["--to-file", "direct.dat", "--regime", "6", "-W", "20e6", "-N", "2", "-r", "9", "--log",
                      "--tx-amplitude", "0.25"])
    ftw_ofdm_tx.main(["--from-file", "ofdm_zerogap.dat", "--to-file", "replay.dat", "--tx-amplitude", "0.25"])
    a, b = np.fromfile("direct.dat", np.complex64), np.fromfile("replay.dat", np.complex64)
    assert a.size == b.size and a.size > 0
    assert np.array_equal(a, b)              # (x * 0.125) * amp is exact either way: 0.125 is a power of two
    del off0
