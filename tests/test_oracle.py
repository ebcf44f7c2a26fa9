"""The oracle itself: pinned against what the reference ships (Annex G.8 frame, literal
tables) and against structural identities; numpy and C restatements must agree bit for bit."""
import zlib

import numpy as np
import pytest

from oracle import ftw_oracle as o

ANNEX_G_BODY = bytes.fromhex(
    "4a6f792c2062726967687420737061726b206f6620646976696e6974792c0a4461756768746572206f6620456c797369756d2c0a"
    "466972652d696e73697265642077652074726561")
# MPDU of ftw_80211_ofdm_tx/src/matlab/encode.m:59-65 (duration 0x002e = TxTime of :150)
ANNEX_G_MPDU = bytes.fromhex("0402" "002e" "006008cd37a6" "0020d6013cf1" "006008ad3baf" "0000") + ANNEX_G_BODY
# FCS bits in transmit order, encode.m:223
ANNEX_G_FCS_BITS = [0, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1]


def annex_g_frame(modulate):
    enc = o.encode_bits_from_mpdu(ANNEX_G_MPDU, "6", o.bits_to_bytes(ANNEX_G_FCS_BITS))
    assert enc["N_sym"] == 6 and len(enc["interleaved"]) == 6 + 6 * 24
    return modulate(enc["interleaved"])


def check_against_annex_g(iq_amp1):
    """encode.m:421-446: reference table = frame / 8; 3-decimal table, boundaries windowed."""
    ref = o.tables().annex_g8_time
    assert len(ref) == 881
    got = np.asarray(iq_amp1)[:880] / 8.0
    d = np.abs(got - ref[:880])
    boundary = np.zeros(880, bool)
    boundary[[0, 160, 320] + list(range(400, 880, 80))] = True
    assert d[~boundary].max() < 1.5e-3          # half an LSB of the table in re and im is 7.1e-4
    assert d[boundary].min() > 0.02             # the table's windowed samples really differ


def test_annex_g8_known_answer_numpy():
    iq = annex_g_frame(lambda m: o.modulate(m, "6", 0, amp=1.0, zerogap=False)["iq"])
    check_against_annex_g(iq)


def test_annex_g8_known_answer_c(c_oracle):
    iq = annex_g_frame(lambda m: c_oracle.modulate(m, 6, 0, amp=1.0, flags=1)["iq64"])
    check_against_annex_g(iq)


def test_txtime_of_annex_g_frame():
    # encode.m:150: 5*4 + 4*ceil((16+8*100+6)/144) + 2 = 46 = 0x2e, the duration in the MPDU above
    assert int(5 * 4.0 + 4.0 * 6) + 2 == 0x2E


def test_crc_identities(c_oracle):
    rng = np.random.default_rng(5)
    for n in (0, 1, 3, 24, 100, 1524, 3164):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = int("{:032b}".format(zlib.crc32(d) & 0xFFFFFFFF)[::-1], 2)
        assert o.ftw_crc32(d) == want == o.ftw_crc32_bitwise(d) == c_oracle.crc32(d)


def test_interleaver_is_a_permutation_and_matches_closed_form():
    for n_cbps, n_bpsc in ((48, 1), (96, 2), (192, 4), (288, 6)):
        p = o.interleave_perm(n_cbps, n_bpsc)
        assert sorted(p.tolist()) == list(range(n_cbps))
        s = max(n_bpsc // 2, 1)
        r_rows = n_cbps // 16
        k = np.arange(n_cbps)
        r, c = k // 16, k % 16
        assert np.array_equal(p, r_rows * c + s * (r // s) + (r - c) % s)
    assert np.array_equal(o.interleave_perm(48, 1, signal=True), 3 * (np.arange(48) % 16) + np.arange(48) // 16)


@pytest.mark.parametrize("row", [
    # payload, regime, N_cbps, N_bpsc, N_dbps, N_sym, N_pad, tail+pad B, PLCP B, msg B, samples, txtime
    (10, "1", 48, 1, 24, 14, 10, 2, 45, 90, 2560, 78), (10, "2", 48, 1, 36, 10, 34, 5, 48, 66, 2240, 62),
    (10, "3", 96, 2, 48, 7, 10, 2, 45, 90, 2000, 50), (10, "4", 96, 2, 72, 5, 34, 5, 48, 66, 1840, 42),
    (10, "5", 192, 4, 96, 4, 58, 8, 51, 102, 1760, 38), (10, "6", 192, 4, 144, 3, 106, 14, 57, 78, 1680, 34),
    (10, "7", 288, 6, 192, 2, 58, 8, 51, 78, 1600, 30), (10, "8", 288, 6, 216, 2, 106, 14, 57, 78, 1600, 30),
    (1500, "1", 48, 1, 24, 511, 18, 3, 1536, 3072, 42320, 2066), (1500, "2", 48, 1, 36, 341, 30, 5, 1538, 2052, 28720, 1386),
    (1500, "3", 96, 2, 48, 256, 42, 6, 1539, 3078, 21920, 1046), (1500, "4", 96, 2, 72, 171, 66, 9, 1542, 2058, 15120, 706),
    (1500, "5", 192, 4, 96, 128, 42, 6, 1539, 3078, 11680, 534), (1500, "6", 192, 4, 144, 86, 138, 18, 1551, 2070, 8320, 366),
    (1500, "7", 288, 6, 192, 64, 42, 6, 1539, 2310, 6560, 278), (1500, "8", 288, 6, 216, 57, 66, 9, 1542, 2058, 6000, 250),
])
def test_frame_geometry(row, c_oracle):
    n, regime, n_cbps, n_bpsc, n_dbps, n_sym, n_pad, tp, plcp, msg, samples, txtime = row
    info = o.get_info(b"x" * n, regime, 4.0)
    assert (info["N_cbps"], info["N_bpsc"], info["N_dbps"], info["N_sym"], info["N_pad"], info["txtime"]) == \
        (n_cbps, n_bpsc, n_dbps, n_sym, n_pad, txtime)
    assert o.tail_and_pad_bytes(n_pad) == tp
    enc = o.encode_bits(b"x" * n, regime)
    assert len(enc["plcp"]) == plcp == len(enc["scrambled"])
    assert len(enc["coded"]) == msg == len(enc["interleaved"])
    assert o.n_samples(n_sym) == samples
    assert c_oracle.packet_sizes(int(regime), n) == {"N_sym": n_sym, "N_pad": n_pad, "plcp_len": plcp, "msg_len": msg}


def test_helloworld_signal_field_and_header():
    enc = o.encode_bits(b"HelloWorld", "1", 4.0)
    plcp = enc["plcp"]
    # RATE 1101, reserved 0, LENGTH 38 = 0b000000100110 LSB first, even parity, tail
    length = 24 + 10 + 4
    bits = "1101" + "0" + "{:012b}".format(length)[::-1]
    bits += str(bits.count("1") % 2) + "000000"
    assert plcp[:3] == int(bits, 2).to_bytes(3, "big")
    assert plcp[3:5] == b"\x00\x00"
    # frame control 08 00 sent LSB first, duration 78 big-endian
    assert plcp[5:9] == bytes([0x10, 0x00, 0x00, int("{:08b}".format(78)[::-1], 2)])
    assert enc["info"]["packet"][2:4] == b"\x00\x4e"


def test_scrambler_leaves_signal_and_zeroes_tail():
    pl = bytes(range(200))
    enc = o.encode_bits(pl, "5")
    a, b = o.bytes_to_bits(enc["plcp"]), o.bytes_to_bits(enc["scrambled"])
    assert np.array_equal(a[:24], b[:24])
    L = enc["Length"]
    assert not b[40 + 8 * L:46 + 8 * L].any()
    seq = o.tables().scrambling_seq
    k = np.arange(len(a) - 24)
    x = a[24:] ^ seq[k % 127]
    x[16 + 8 * L:22 + 8 * L] = 0
    assert np.array_equal(x, b[24:])


def test_puncture_keeps_expected_positions():
    rng = np.random.default_rng(2)
    pl = rng.integers(0, 256, 77, dtype=np.uint8).tobytes()
    full = o.bytes_to_bits(o.encode_bits(pl, "5")["coded"])      # rate 1/2, 16-QAM
    for regime, keep in (("6", lambda p: np.isin(p % 6, (0, 1, 2, 5))), ):
        enc = o.encode_bits(pl, regime)
        # re-encode the scrambled stream at rate 1/2 by hand
        u = o.bytes_to_bits(enc["scrambled"])
        up = np.concatenate([np.zeros(6, np.uint8), u])
        d = lambda k: up[6 - k:6 - k + len(u)]
        a = d(0) ^ d(2) ^ d(3) ^ d(5) ^ d(6)
        b = d(0) ^ d(1) ^ d(2) ^ d(3) ^ d(6)
        c = np.empty(2 * len(u), np.uint8)
        c[0::2], c[1::2] = a, b
        data = c[48:]
        kept = data[keep(np.arange(len(data)))][:enc["info"]["N_sym"] * 192]
        got = o.bytes_to_bits(enc["coded"])
        assert np.array_equal(got[:48], c[:48]) and np.array_equal(got[48:], kept)
    assert len(full) == 48 + o.get_info(pl, "5", 4.0)["N_sym"] * 192


def test_c_and_numpy_restatements_agree(c_oracle):
    rng = np.random.default_rng(0)
    for regime in "12345678":
        for n in (0, 1, 10, 100, 1500, 2304, 3140):
            pl = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            a = o.tx_packet(pl, regime, 4.0, pilot_offset=37)
            b = c_oracle.encode_bits(pl, int(regime), 4.0)
            for key in ("plcp", "scrambled", "coded", "interleaved"):
                assert a[key] == b[key], (regime, n, key)
            m = c_oracle.modulate(b["interleaved"], int(regime), 37)
            assert np.array_equal(m["mapped"], a["mapped"])
            assert np.abs(m["iq64"] - a["iq"]).max() < 1e-12
            assert np.abs(m["iq"].astype(np.complex128) - a["iq"]).max() < 1e-6


def test_stream_pilot_counter_runs_across_packets():
    p = [b"abc", b"defgh"]
    s = o.tx_stream(p, "3")
    first = o.tx_packet(p[0], "3", pilot_offset=0)
    second = o.tx_packet(p[1], "3", pilot_offset=first["info"]["N_sym"] + 1)
    assert np.array_equal(s, np.concatenate([first["iq"], second["iq"]]))
    restart = o.tx_packet(p[1], "3", pilot_offset=0)
    assert not np.array_equal(second["iq"], restart["iq"])


def test_errors_follow_the_reference():
    with pytest.raises(ValueError):
        o.ftw_make(b"x" * 3141, "1", 4.0)        # packet_len 3169 > 3168 (ftw_packet_utils.py:119-121)
    o.ftw_make(b"x" * 3140, "1", 4.0)
    with pytest.raises(KeyError):
        o.get_info(b"", "9", 4.0)                # reference: UnboundLocalError (no else branch :39-78)


# ---- third restatement (per-bit Python, reference-shaped control flow) against the numpy oracle

@pytest.mark.parametrize("regime", [str(r) for r in range(1, 9)])
def test_per_bit_restatement_agrees(regime):
    from oracle import ftw_pybits as pb
    rng = np.random.default_rng(40 + int(regime))
    for n, t in ((0, 4.0), (1, 4.0), (37, 8.0), (100, 4.0), (333, 160.0)):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        a = pb.encode_bits(payload, regime, t)
        b = o.encode_bits(payload, regime, t)
        for k in ("plcp", "scrambled", "coded", "interleaved"):
            assert a[k] == b[k], (k, n)
        assert a["Length"] == b["Length"] and a["N_sym"] == b["info"]["N_sym"]


def test_per_bit_lfsr_is_the_reference_sequence():
    from oracle import ftw_pybits as pb
    assert pb.lfsr_sequence() == list(o.tables().scrambling_seq)


# ---- a receiver written from the 802.11a standard decodes what the oracle transmits, at every rate

@pytest.mark.parametrize("regime", [str(r) for r in range(1, 9)])
def test_standard_receiver_decodes_oracle_frames(regime):
    import rx80211a as rx
    rng = np.random.default_rng(300 + int(regime))
    for n in (0, 61):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        r = o.tx_packet(payload, regime, 4.0, pilot_offset=5, amp=0.3)
        got = rx.receive(r["iq"], gain=0.3)
        assert got["length"] == 24 + n + 4 and got["n_sym"] == r["info"]["N_sym"]
        assert got["rate_bits"] == {"1": 0xD, "2": 0xF, "3": 0x5, "4": 0x7, "5": 0x9, "6": 0xB, "7": 0x1, "8": 0x3}[regime]
        assert not got["service"].any()
        assert got["fcs_ok"], "FCS of the decoded PSDU"
        assert got["psdu"][:24] == r["info"]["packet"][:24] and got["psdu"][24:-4] == payload


# ---- PNC roles (ftw_pnc_ofdm_preamble.cc:80-103, ftw_pnc_ofdm_pilot_cc.cc:125-178, ftw_pnc_zerogap.cc:81-92)

@pytest.mark.parametrize("role", ["A", "B"])
def test_pnc_role_frames(role, c_oracle):
    """Structure of a PNC frame, checked on the samples: six preamble rows with the long-training
    pair before (A) or after (B) two zero rows; constant pilots 1,0,1,0 / 0,1,0,1 at -21,-7,+7,+21
    on every symbol, SIGNAL included; 13 zero rows behind; the data carriers are those of the
    standard frame.  The C and numpy restatements agree."""
    import rx80211a as rx
    rng = np.random.default_rng(77)
    payload = rng.integers(0, 256, 200, dtype=np.uint8).tobytes()
    for regime in ("1", "6", "8"):
        std = o.tx_packet(payload, regime, 4.0, pilot_offset=9, amp=0.3)
        pnc = o.tx_packet(payload, regime, 4.0, pilot_offset=9, amp=0.3, role=role)
        n_sym = std["info"]["N_sym"]
        rows = pnc["iq"].reshape(-1, 80)
        srows = std["iq"].reshape(-1, 80)
        assert rows.shape[0] == 6 + 1 + n_sym + 13 and srows.shape[0] == 4 + 1 + n_sym + 13
        lts, zero = (slice(2, 4), slice(4, 6)) if role == "A" else (slice(4, 6), slice(2, 4))
        assert np.array_equal(rows[0:2], srows[0:2]) and np.array_equal(rows[lts], srows[2:4])
        assert not rows[zero].any() and not rows[-13:].any()
        spec = np.fft.fft(rows[6:6 + 1 + n_sym, 16:], axis=1) / (8 * float(np.float32(0.3)))
        sspec = np.fft.fft(srows[4:4 + 1 + n_sym, 16:], axis=1) / (8 * float(np.float32(0.3)))
        want = {"A": (1, 0, 1, 0), "B": (0, 1, 0, 1)}[role]
        for f, w in zip((-21, -7, 7, 21), want):
            assert np.allclose(spec[:, f % 64], w, atol=1e-9)
        data = [f % 64 for f in rx.DATA_FREQS]
        assert np.allclose(spec[:, data], sspec[:, data], atol=1e-9)
        # a standard receiver skips the two extra rows and decodes the frame (it does not use pilots)
        got = rx.receive(pnc["iq"][160:], gain=0.3)
        assert got["fcs_ok"] and got["psdu"][24:-4] == payload
        flags = 3 | (8 if role == "A" else 16) | ((2 if role == "A" else 3) << 8)
        c = c_oracle.modulate(std["interleaved"], int(regime), 9, 0.3, flags)
        assert np.allclose(c["iq64"], pnc["iq"], atol=1e-12)


def test_pnc_pilot_block_user_0():
    """user 0 fills all four pilots with +1 (ftw_pnc_ofdm_pilot_cc.cc:130-131); any user other
    than 0 and 1 behaves as user 2 (:127)."""
    sym = np.zeros((3, 48), np.complex64)
    for user, want in ((0, (1, 1, 1, 1)), (1, (1, 0, 1, 0)), (2, (0, 1, 0, 1)), (7, (0, 1, 0, 1))):
        p = o.ofdm_pilot(sym, 0, user)
        assert [p[1, i].real for i in (5, 19, 33, 47)] == list(want)
        assert not np.delete(p, (5, 19, 33, 47), axis=1).any()


def test_psdu_entry_equals_the_mpdu_entry():
    """encode_bits_from_psdu (what raw-PSDU plans are checked against) is encode_bits_from_mpdu with the
    FCS taken from the PSDU's last four bytes, for the Annex G frame and for random ones; and for an
    ordinary packet with its own header and CRC it is encode_bits."""
    fcs_tx = o.bits_to_bytes(np.array(ANNEX_G_FCS_BITS, np.uint8))
    psdu = ANNEX_G_MPDU + o._bitrev8(fcs_tx)
    assert o.encode_bits_from_psdu(psdu, "6") == o.encode_bits_from_mpdu(ANNEX_G_MPDU, "6", fcs_tx)
    rng = np.random.default_rng(5)
    for regime in "12345678":
        n = int(rng.integers(4, 400))
        p = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert o.encode_bits_from_psdu(p, regime) == o.encode_bits_from_mpdu(p[:-4], regime, o._bitrev8(p[-4:]))
        payload = rng.integers(0, 256, 77, dtype=np.uint8).tobytes()
        full = o.encode_bits(payload, regime)
        mpdu = o.get_info(payload, regime, 4.0)["packet"]
        whole = mpdu + o._bitrev8(o.ftw_crc32(mpdu).to_bytes(4, "big"))
        raw = o.encode_bits_from_psdu(whole, regime)
        assert all(raw[k] == full[k] for k in ("plcp", "scrambled", "coded", "interleaved"))
    short = o.encode_bits_from_psdu(b"\x5a", "1")        # fewer than four bytes: no FCS to speak of
    assert short["Length"] == 1 and short["N_sym"] == 2 and short["plcp"][3:6] == b"\x00\x00\x5a"
