"""Variable-size modulator (rawofdm chain): oracle self-checks on the CPU, GPU parity against the
oracle, and the loop-back property the reference's own test asserts
(rawofdm/src/python/qa_ofdm.py:40-67: unit-power QPSK in, SNR > 40 dB after demodulation).

PARITY UNPINNED by reference vectors: the reference holds no golden output for this chain, so
the oracle (oracle/raw_oracle.py) is a restatement checked by structure and loop-back only.
IQ tolerance: 1e-5 relative L2 per OFDM symbol row (fp32 against the float64 oracle); the mapper
tap is compared bit for bit.
"""
import numpy as np
import pytest

from oracle import raw_oracle as ro

REL_L2_TOL = 1e-5


def qpsk(rng, n):
    return np.exp(1j * (np.pi / 4) * (2 * rng.integers(0, 4, n) + 1)).astype(np.complex64)


def ideal_demod(params, rows):
    """Strip the cyclic prefix, forward FFT, undo the DC-centre shift, pick the data carriers."""
    n, cp = params.fft_length, params.cp_length
    t = np.asarray(rows, dtype=np.complex128).reshape(-1, n + cp)[:, cp:]
    f = np.fft.fft(t, axis=1) / np.sqrt(n)
    f = np.roll(f, n // 2, axis=1)
    return f[:, np.asarray(params.padded_carriers) == 1]


# ------------------------------------------------------------------------------ CPU: oracle

def test_params_default_geometry():
    p = ro.OfdmParams(ro.Options())                       # qa_ofdm.py:16-24 option set
    assert (p.fft_length, p.occupied_tones, p.cp_length) == (64, 53, 16)
    assert p.pilot_tones == 4 and p.data_tones == 48 and p.pad == 6
    cm = np.asarray(p.padded_carriers)
    assert len(cm) == 64 and cm[32] == 0                  # DC in the middle (raw_ofdm_params.py:20-22)
    assert np.array_equal(cm[7:32], cm[33:58][::-1])      # mirrored about DC
    assert list(np.flatnonzero(cm == 2)) == [6 + 5, 6 + 18, 32 + 26 - 18, 32 + 26 - 5]
    assert len(p.padded_preambles) == 3
    pre = np.asarray(p.padded_preambles)
    assert np.all(pre[:, 32] == 0) and np.all(pre[:, :6] == 0) and np.all(pre[:, 59:] == 0)
    assert np.all(pre[0, 1::2] == 0)                      # half-sync: even absolute bins only
    assert np.allclose(np.abs(pre[0, 6:59:2][np.arange(27) != 13]), np.sqrt(2))
    assert np.allclose(np.abs(pre[1:, 6:59][:, np.arange(53) != 26]), 1.0)


def test_params_full_sync_repeats_first_preamble():
    p = ro.OfdmParams(ro.Options(half_sync=False))
    assert len(p.padded_preambles) == 4 and p.padded_preambles[0] == p.padded_preambles[1]


def test_params_product_mirror_matches_oracle():
    from ofdm_amd import raw
    for kw in (dict(), dict(fft_length=256, occupied_tones=208, cp_length=64),
               dict(fft_length=1024, occupied_tones=832, cp_length=128, num_preambles=1, half_sync=False),
               dict(fft_length=63, occupied_tones=51)):
        a, b = ro.OfdmParams(ro.Options(**kw)), raw.ofdm_params(ro.Options(**kw))
        for k in ("fft_length", "occupied_tones", "cp_length", "pad", "data_tones", "pilot_tones",
                  "padded_carriers", "padded_preambles", "carriers"):
            assert getattr(a, k) == getattr(b, k), k


@pytest.mark.parametrize("kw", [dict(), dict(fft_length=256, occupied_tones=208, cp_length=64),
                                dict(fft_length=1024, occupied_tones=832, cp_length=128, num_preambles=1, half_sync=False),
                                dict(fft_length=63, occupied_tones=51)])
def test_params_product_structure_from_the_reference_text(kw):
    """The product's ofdm_params against what rawofdm/src/python/raw_ofdm_params.py states, without
    the oracle in between: even-rounded sizes (:52-53), DC null at fft_length/2 (:20-22,36-37),
    mirror symmetry (:88), pilots at range(5, half, 13) from the band edge (:81-85), unit-modulus
    preamble values from the generator seeded with 9817 in carrier order (:29-37,61-62), even-bin rule
    with sqrt(2) for the half-sync symbol (:39-48), first symbol repeated otherwise (:71-73)."""
    import cmath
    import math
    import random
    from ofdm_amd import raw
    opt = ro.Options(**kw)
    p = raw.ofdm_params(opt)
    n = opt.fft_length + (opt.fft_length & 1)
    occ = opt.occupied_tones + (opt.occupied_tones & 1)
    half = occ // 2
    assert (p.fft_length, p.occupied_tones, p.pad) == (n, occ + 1, (n - occ) // 2)
    dc = p.pad + half
    assert dc == n // 2
    cm = np.asarray(p.padded_carriers)
    assert cm.shape == (n,) and cm[dc] == 0
    assert np.all(cm[:p.pad] == 0) and np.all(cm[dc + half + 1:] == 0)
    assert np.array_equal(cm[p.pad:dc], cm[dc + 1:dc + half + 1][::-1])
    assert list(np.flatnonzero(cm[p.pad:dc] == 2)) == list(range(5, half, 13))
    assert set(cm[p.pad:dc].tolist()) <= {1, 2}
    assert p.pilot_tones == 2 * len(range(5, half, 13)) and p.data_tones == occ - p.pilot_tones
    assert p.carriers == cm[p.pad:dc + half + 1].tolist()
    # preamble values: the generator's stream, one value per occupied carrier, DC skipped
    random.seed(9817)
    n_sym = opt.num_preambles + 1
    want = np.zeros((n_sym, n), complex)
    for s_ in range(n_sym):
        for c in list(range(p.pad, dc)) + list(range(dc + 1, dc + half + 1)):
            want[s_, c] = cmath.exp(1j * 2 * math.pi * random.random())
    if opt.half_sync:
        want[0, 1::2] = 0
        want[0, 0::2] *= math.sqrt(2)
    else:
        want = np.concatenate([want[:1], want])
    got = np.asarray(p.padded_preambles)
    assert got.shape == want.shape and np.array_equal(got, want)
    occupied = np.abs(want[-1]) > 0
    assert np.allclose(np.abs(got[-1][occupied]), 1.0) and occupied.sum() == occ
    assert np.asarray(p.preambles).shape == (n_sym, occ + 1)


def test_survey_sizes():
    p = ro.OfdmParams(ro.Options(fft_length=256, occupied_tones=208, cp_length=64))
    assert (p.pilot_tones, p.data_tones) == (16, 192)
    p = ro.OfdmParams(ro.Options(fft_length=1024, occupied_tones=832, cp_length=256))
    assert (p.pilot_tones, p.data_tones) == (64, 768)


def test_mapper_rejects_bad_map():
    with pytest.raises(ValueError):
        ro.mapper([0, 1, 3, 1], np.zeros((1, 2)))


def test_insert_preamble_drops_leading_symbols():
    sym = np.arange(5 * 4).reshape(5, 4).astype(np.complex64)
    pre = np.full((2, 4), -1, np.complex64)
    out = ro.insert_preamble(sym, [0, 1, 0, 1, 0], pre)
    assert out.shape == (8, 4)
    assert np.array_equal(out[2], sym[1]) and np.array_equal(out[3], sym[2]) and np.array_equal(out[6], sym[3])


def test_oracle_loopback():
    rng = np.random.default_rng(4)
    p = ro.OfdmParams(ro.Options(fft_length=256, occupied_tones=208, cp_length=64))
    data = qpsk(rng, 8 * p.data_tones).reshape(8, -1)
    rows = ro.modulate(p, data, ro.frame_flags(2, 4))
    assert rows.shape == (2 * (3 + 4), 320)
    rx = ideal_demod(p, rows).reshape(2, 7, -1)[:, 3:].reshape(8, -1)
    mse = np.mean(np.abs(rx - data) ** 2)
    assert 10 * np.log10(1.0 / mse) > 100


# ------------------------------------------------------------------------------ GPU parity

def rel_l2(got, ref, row_len):
    got = np.asarray(got).reshape(-1, row_len).astype(np.complex128)
    ref = np.asarray(ref).reshape(-1, row_len)
    assert got.shape == ref.shape
    return np.sqrt((np.abs(got - ref) ** 2).sum(1)) / np.sqrt((np.abs(ref) ** 2).sum(1))


CASES = [
    dict(fft_length=64, occupied_tones=52, cp_length=16),
    dict(fft_length=128, occupied_tones=104, cp_length=32),
    dict(fft_length=256, occupied_tones=208, cp_length=64),
    dict(fft_length=256, occupied_tones=208, cp_length=16, num_preambles=1, half_sync=False),
    dict(fft_length=512, occupied_tones=416, cp_length=64),
    dict(fft_length=1024, occupied_tones=832, cp_length=256),
    dict(fft_length=1024, occupied_tones=832, cp_length=128, num_preambles=3),
    dict(fft_length=1024, occupied_tones=832, cp_length=16),
    dict(fft_length=256, occupied_tones=100, cp_length=7),           # odd row length, narrow band
    dict(fft_length=64, occupied_tones=52, cp_length=0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CASES, ids=lambda k: "N%d_cp%d" % (k["fft_length"], k["cp_length"]))
def test_gpu_frames_match_oracle(txctx, kw):
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(4)
    opt = ro.Options(**kw)
    mod = raw.ofdm_mod(opt, ctx=txctx)
    p = ro.OfdmParams(opt)
    n_frames, size = 5, 16
    data = (rng.standard_normal((n_frames * size, p.data_tones)) +
            1j * rng.standard_normal((n_frames * size, p.data_tones))).astype(np.complex64)
    ref = ro.modulate(p, data, ro.frame_flags(n_frames, size))
    plan = raw.plan_frames(txctx, p.fft_length, p.cp_length, p.padded_carriers, p.padded_preambles, n_frames, size)
    assert plan.data_tones == p.data_tones and plan.n_rows == ref.shape[0]
    iq, mapped = plan.modulate(torch.from_numpy(data).to(txctx.device).reshape(-1), want_mapped=True)
    torch.cuda.synchronize()
    err = rel_l2(iq.cpu().numpy(), ref, p.fft_length + p.cp_length)
    assert err.max() <= REL_L2_TOL, err.max()
    want = ro.mapper(p.padded_carriers, data)
    assert np.array_equal(mapped.cpu().numpy().view(np.uint64), want.view(np.uint64)), "mapper tap"
    # the drop-in class gives the same stream
    iq2 = mod.modulate_frames(data, size)
    torch.cuda.synchronize()
    assert np.array_equal(iq2.cpu().numpy().view(np.uint64), iq.cpu().numpy().view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 256, 1024])
def test_gpu_flags_match_oracle(txctx, n):
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(5)
    opt = ro.Options(fft_length=n, occupied_tones=int(0.8125 * n), cp_length=n // 8)
    p = ro.OfdmParams(opt)
    flags = np.zeros(37, np.uint8)
    flags[[3, 4, 20, 36]] = 1                      # three symbols dropped in front, ragged frames, a frame of one
    flags[10] = 2                                  # only bit 0 counts
    data = qpsk(rng, 37 * p.data_tones).reshape(37, -1)
    ref = ro.modulate(p, data, flags)
    mod = raw.ofdm_mod(opt, ctx=txctx)
    iq = mod.modulate(data, flags)
    torch.cuda.synchronize()
    assert iq.numel() == ref.size
    err = rel_l2(iq.cpu().numpy(), ref, n + n // 8)
    assert err.max() <= REL_L2_TOL, err.max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,cp", [(64, 16), (256, 7), (512, 128), (1024, 256)])
def test_gpu_log_taps_match_oracle(txctx, tmp_path, monkeypatch, n, cp):
    """tx-map / tx-pre / tx-ifft / tx-cp of ofdm_mod's --log (rawofdm/src/python/raw_ofdm_mod.py:54-66):
    item sizes and contents against the oracle's intermediate results, ragged frames with dropped symbols."""
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(15)
    opt = ro.Options(fft_length=n, occupied_tones=int(0.8125 * n), cp_length=cp, log=True)
    p = ro.OfdmParams(opt)
    flags = np.zeros(23, np.uint8)
    flags[[2, 9, 22]] = 1
    data = qpsk(rng, 23 * p.data_tones).reshape(23, -1)
    mapped = ro.mapper(p.padded_carriers, data)
    with_pre = ro.insert_preamble(mapped, flags, p.padded_preambles)
    rows = ro.ifft_cp_scale(with_pre, cp) * np.sqrt(n)               # in front of the 1/sqrt(N) block
    monkeypatch.chdir(tmp_path)
    mod = raw.ofdm_mod(opt, ctx=txctx)
    iq = mod.modulate(data, flags)
    torch.cuda.synchronize()
    assert iq.numel() == rows.size
    got = {k: np.fromfile(str(tmp_path / ("tx-%s.dat" % k)), dtype=np.complex64) for k in ("map", "pre", "ifft", "cp")}
    assert np.array_equal(got["map"].view(np.uint64), mapped.astype(np.complex64).reshape(-1).view(np.uint64))
    assert np.array_equal(got["pre"].view(np.uint64), with_pre.astype(np.complex64).reshape(-1).view(np.uint64))
    assert got["ifft"].size == rows.shape[0] * n and got["cp"].size == rows.size
    assert rel_l2(got["cp"], rows, n + cp).max() <= REL_L2_TOL
    assert rel_l2(got["ifft"], rows[:, cp:], n).max() <= REL_L2_TOL
    # frames of equal size take the arithmetic row mapping
    plan = raw.plan_frames(txctx, n, cp, p.padded_carriers, p.padded_preambles, 3, 4)
    d2 = torch.from_numpy(data[:12].astype(np.complex64)).to(txctx.device).reshape(-1)
    iq2, m2 = plan.modulate(d2, want_mapped=True)
    taps = plan.taps(m2, iq2)
    torch.cuda.synchronize()
    want = ro.insert_preamble(mapped[:12], ro.frame_flags(3, 4), p.padded_preambles).astype(np.complex64)
    assert np.array_equal(taps["pre"].cpu().numpy().view(np.uint64), want.view(np.uint64))
    assert torch.equal((taps["ifft"].reshape(-1, n)).view(torch.float32),
                       taps["cp"].reshape(-1, n + cp)[:, cp:].contiguous().view(torch.float32))


@pytest.mark.gpu
def test_gpu_no_flag_no_output(txctx):
    from ofdm_amd import raw
    p = ro.OfdmParams(ro.Options())
    plan = raw.plan_flags(txctx, 64, 16, p.padded_carriers, p.padded_preambles, np.zeros(9, np.uint8))
    assert plan.n_rows == 0 and plan.n_frames == 0


@pytest.mark.gpu
def test_gpu_rejects_bad_shapes(txctx):
    from ofdm_amd import _lib, raw
    p = ro.OfdmParams(ro.Options())
    bad = list(p.padded_carriers)
    bad[10] = 3
    with pytest.raises(_lib.OfdmTxError) as e:     # raw_ofdm_mapper.cc:37-38 throws std::invalid_argument
        raw.plan_frames(txctx, 64, 16, bad, p.padded_preambles, 1, 4)
    assert e.value.code == _lib.ESHAPE
    with pytest.raises(_lib.OfdmTxError) as e:
        raw.plan_frames(txctx, 96, 16, [1] * 96, np.zeros((1, 96)), 1, 4)
    assert e.value.code == _lib.ESHAPE


@pytest.mark.gpu
@pytest.mark.parametrize("n,cp", [(256, 64), (1024, 128)])
def test_gpu_loopback_large(txctx, n, cp):
    """qa_ofdm.py:40-67 at batch size: unit-power QPSK, SNR after an ideal receiver > 40 dB
    (here > 100 dB: no channel, fp32 transform)."""
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(6)
    opt = ro.Options(fft_length=n, occupied_tones=int(0.8125 * n), cp_length=cp)
    p = ro.OfdmParams(opt)
    size, n_frames = 16, (1 << 16) // 16 if n == 256 else (1 << 13) // 16
    data = qpsk(rng, n_frames * size * p.data_tones)
    assert abs(np.mean(np.abs(data) ** 2) - 1.0) < 1e-6
    mod = raw.ofdm_mod(opt, ctx=txctx)
    iq = mod.modulate_frames(data, size).cpu().numpy()
    rows = iq.reshape(n_frames, 3 + size, n + cp)
    rx = ideal_demod(p, rows[:, 3:].reshape(-1, n + cp)).reshape(-1)
    mse = np.mean(np.abs(rx - data) ** 2)
    assert 10 * np.log10(1.0 / mse) > 100
    # every frame starts with the same preamble rows
    assert np.array_equal(rows[:, :3].view(np.uint64), np.broadcast_to(rows[:1, :3], rows[:, :3].shape).view(np.uint64))
    # average transmit power per sample ~ occupied fraction of the band
    pw = np.mean(np.abs(rows[:, 3:]) ** 2)
    assert abs(pw - (p.occupied_tones - 1) / n) < 0.01


# ------------------------------------------------------------------------------ TX bit chain, CPU

@pytest.mark.parametrize("nbits", [1, 2, 4, 6])
def test_oracle_intrlv_roundtrip(nbits):
    """rawofdm/src/python/qa_intrlv.py:19-40."""
    rng = np.random.default_rng(nbits)
    data = rng.integers(0, 2, 10 * 48 * nbits).astype(np.uint8)
    enc = ro.intrlv_bit(data, 48, nbits, False)
    assert np.array_equal(ro.intrlv_bit(enc, 48, nbits, True), data)
    assert sorted(ro.intrlv_bit_map(48, nbits)) == list(range(48 * nbits))


@pytest.mark.parametrize("nbits", [1, 2, 4, 6])
def test_oracle_qam_roundtrip_and_power(nbits):
    """rawofdm/src/python/qa_qam.py:16-52: slicing the decoder's confidences at 128 returns the
    bits; average symbol power is 1."""
    rng = np.random.default_rng(10 + nbits)
    bits = rng.integers(0, 2, 6000 * nbits).astype(np.uint8)
    sym = ro.qam_enc(bits, nbits)
    dec = ro.qam_dec(sym, nbits)
    assert np.array_equal((dec.astype(int) - 128 >= 0).astype(np.uint8), bits)
    allbits = np.array([[(v >> (nbits - 1 - i)) & 1 for i in range(nbits)] for v in range(1 << nbits)]).reshape(-1)
    assert abs(np.mean(np.abs(ro.qam_enc(allbits, nbits)) ** 2) - 1.0) < 1e-6


@pytest.mark.parametrize("nc,np_", [(4, 4), (6, 4), (4, 3)])
def test_oracle_conv_roundtrip(nc, np_):
    """rawofdm/src/python/qa_conv.py:37-105 at a smaller length: puncture, de-puncture with
    erasures, decode = identity (last byte is the zero padding)."""
    rng = np.random.default_rng(nc * 10 + np_)
    data = bytearray(rng.integers(0, 256, 96, dtype=np.uint8).tobytes())
    data[-1] = 0
    coded, _ = ro.conv_enc(data)
    punct = ro.conv_punc(coded, nc, np_)
    assert len(punct) == len(data) * 8 * 2 * np_ // nc
    soft = ro.conv_punc(punct * 255, np_, nc, 128)
    assert ro.viterbi_decode(soft, len(data) * 8) == bytes(data)


@pytest.mark.parametrize("bitrate", range(1, 9))
def test_oracle_fullstack_noiseless(bitrate):
    """rawofdm/src/python/qa_qamconv.py:36-133 without the noise source: 14 symbols x 48
    carriers per frame, several frames, received bytes = sent bytes."""
    rng = np.random.default_rng(bitrate)
    qambits, punc = ro.BITRATES[bitrate]
    fb = ro.framebytes((qambits, punc), 48 * 16) - 1
    data = rng.integers(0, 256, 3 * fb, dtype=np.uint8).tobytes()
    values, taps = ro.qam_tx(bitrate, 48, 16, data)
    assert len(values) == 3 * 48 * 16
    assert ro.qam_rx(bitrate, 48, 16, values) == data


# ------------------------------------------------------------------------------ TX bit chain, GPU

@pytest.mark.gpu
@pytest.mark.parametrize("bitrate", range(1, 9))
@pytest.mark.parametrize("ncarriers,nsymbols", [(48, 16), (192, 16), (768, 4)])
def test_gpu_qam_tx_bit_exact(txctx, bitrate, ncarriers, nsymbols):
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(100 * bitrate + ncarriers)
    tx = raw.qam_tx(bitrate, ncarriers, nsymbols, ctx=txctx)
    qambits, punc = ro.BITRATES[bitrate]
    assert tx.framebytes == ro.framebytes((qambits, punc), ncarriers * nsymbols) - 1
    data = rng.integers(0, 256, 7 * tx.framebytes, dtype=np.uint8)
    got = tx.modulate(torch.from_numpy(data)).cpu().numpy()
    want, _ = ro.qam_tx(bitrate, ncarriers, nsymbols, data.tobytes())
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate", range(3, 9))
@pytest.mark.parametrize("ncarriers,nsymbols,n_frames", [(40, 16, 5), (24, 16, 1), (8, 32, 9), (56, 8, 300)])
def test_gpu_qam_tx_carrier_counts_off_the_16_grid(txctx, bitrate, ncarriers, nsymbols, n_frames):
    """Carrier counts that are not multiples of 16 take the per-carrier form of the kernel (a
    carrier's bits then need not share one bit position of the interleaver's 16 columns); many
    frames make several passes per workgroup; one frame has both buffer ends in its first words."""
    import torch
    from ofdm_amd import raw
    qambits, punc = ro.BITRATES[bitrate]
    rng = np.random.default_rng(1000 * bitrate + ncarriers)
    tx = raw.qam_tx(bitrate, ncarriers, nsymbols, ctx=txctx)
    data = rng.integers(0, 256, n_frames * tx.framebytes, dtype=np.uint8)
    got = tx.modulate(torch.from_numpy(data)).cpu().numpy()
    want, _ = ro.qam_tx(bitrate, ncarriers, nsymbols, data.tobytes())
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
def test_gpu_qam_tx_many_frames(txctx):
    """Enough frames that every workgroup makes several passes; odd frame sizes (words that
    straddle frame boundaries).  Frames are independent, so the oracle runs on slices of them:
    the first, the last and some in the middle."""
    import torch
    from ofdm_amd import raw
    rng = np.random.default_rng(5)
    for bitrate, ncarriers, nsymbols, n_frames in ((8, 192, 16, 6000), (7, 48, 16, 12000), (2, 208, 4, 20000),
                                                   (5, 768, 2, 9000), (6, 40, 16, 9000)):
        tx = raw.qam_tx(bitrate, ncarriers, nsymbols, ctx=txctx)
        fb, per_frame = tx.framebytes, ncarriers * nsymbols
        data = rng.integers(0, 256, n_frames * fb, dtype=np.uint8)
        got = tx.modulate(torch.from_numpy(data)).cpu().numpy()
        assert got.size == n_frames * per_frame
        for f0, f1 in ((0, 6), (n_frames // 2 - 3, n_frames // 2 + 3), (n_frames - 6, n_frames), (1234, 1237)):
            want, _ = ro.qam_tx(bitrate, ncarriers, nsymbols, data[f0 * fb:f1 * fb].tobytes())
            assert np.array_equal(got[f0 * per_frame:f1 * per_frame].view(np.uint64), want.view(np.uint64)), \
                (bitrate, ncarriers, f0)
        # every value is a constellation point of the right power on average (raw_qam.cc:39-52)
        assert abs(float(np.mean(np.abs(got) ** 2)) - 1.0) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate,nfft", [(1, 64), (4, 256), (6, 256), (7, 1024), (8, 1024)])
def test_gpu_bytes_to_iq_and_back(txctx, bitrate, nfft):
    """Bytes -> qam_tx -> ofdm_mod on the GPU; an ideal receiver (FFT, de-map, soft decisions,
    de-interleave, de-puncture, maximum-likelihood decoder) returns the bytes: the noiseless case of
    rawofdm/src/python/qa_qamconv.py joined with qa_ofdm.py."""
    from ofdm_amd import raw
    rng = np.random.default_rng(bitrate)
    opt = ro.Options(fft_length=nfft, occupied_tones=int(0.8125 * nfft), cp_length=nfft // 4)
    p = ro.OfdmParams(opt)
    nsymbols, n_frames = 16, 3
    tx = raw.qam_tx(bitrate, p.data_tones, nsymbols, ctx=txctx)
    data = rng.integers(0, 256, n_frames * tx.framebytes, dtype=np.uint8).tobytes()
    values = tx.modulate(data)
    iq = raw.ofdm_mod(opt, ctx=txctx).modulate_frames(values, nsymbols).cpu().numpy()
    rows = iq.reshape(n_frames, 3 + nsymbols, -1)[:, 3:].reshape(n_frames * nsymbols, -1)
    rx = ideal_demod(p, rows).reshape(-1).astype(np.complex64)
    assert ro.qam_rx(bitrate, p.data_tones, nsymbols, rx) == data


@pytest.mark.gpu
@pytest.mark.parametrize("nfft,cp,frame", [(64, 16, 16), (64, 16, 4), (128, 32, 16), (256, 64, 16), (256, 7, 16), (256, 64, 5),
                                           (256, 64, 1), (512, 64, 16), (1024, 256, 4), (1024, 16, 4)])
def test_gpu_bytes_to_iq_in_one_kernel(txctx, nfft, cp, frame):
    """ofdm_mod.modulate_bytes (one kernel: bit chain + modulator) gives exactly the samples of
    qam_tx followed by modulate_frames, at every rate; frames enough for several passes."""
    import torch
    from ofdm_amd import raw
    opt = ro.Options(fft_length=nfft, occupied_tones=int(0.8125 * nfft), cp_length=cp)
    mod = raw.ofdm_mod(opt, ctx=txctx)
    p = mod.params
    assert p.data_tones % 16 == 0
    rng = np.random.default_rng(nfft + cp)
    for bitrate in range(1, 9):
        nsymbols = frame                 # (frames shorter than a wave's pass: several frames start inside one pass)
        n_frames = (700 if nfft <= 128 else 150) * 16 // frame + 1
        tx = raw.qam_tx(bitrate, p.data_tones, nsymbols, ctx=txctx)
        data = torch.from_numpy(rng.integers(0, 256, n_frames * tx.framebytes, dtype=np.uint8)).to(txctx.device)
        mod.fused = False
        got = mod.modulate_bytes(bitrate, data, nsymbols)
        # one kernel at every size; the rate (1, (6, 4)) -- "DONT USE" in raw_qam.py:21 -- gives a group of 16
        # carriers a byte and a half of input and is left to the two kernels
        assert mod.fused == (bitrate != 2), (bitrate, nfft)
        want = mod.modulate_frames(tx.modulate(data), nsymbols)
        assert torch.equal(got.view(torch.float32), want.view(torch.float32)), (bitrate, nfft)


@pytest.mark.gpu
def test_gpu_bytes_to_iq_falls_back(txctx):
    """A carrier count off the 16 grid: modulate_bytes runs the two kernels and says so."""
    import torch
    from ofdm_amd import raw
    opt = ro.Options(fft_length=64, occupied_tones=44, cp_length=16)
    mod = raw.ofdm_mod(opt, ctx=txctx)
    p = mod.params
    assert p.data_tones == 40
    tx = raw.qam_tx(5, p.data_tones, 8, ctx=txctx)
    data = torch.from_numpy(np.random.default_rng(1).integers(0, 256, 9 * tx.framebytes, dtype=np.uint8)).to(txctx.device)
    mod.fused = False
    got = mod.modulate_bytes(5, data, 8)
    assert not mod.fused
    assert torch.equal(got.view(torch.float32), mod.modulate_frames(tx.modulate(data), 8).view(torch.float32))


@pytest.mark.gpu
def test_gpu_qam_tx_rejects_bad_geometry(txctx):
    from ofdm_amd import _lib, raw
    with pytest.raises(AssertionError):            # raw_qam.py:35
        raw.qam_tx(8, 50, 3, ctx=txctx)
    tx = raw.qam_tx(8, 48, 16, ctx=txctx)
    with pytest.raises(ValueError):
        tx.modulate(b"\x00" * (tx.framebytes + 1))
    tx.ncarriers, tx.nsymbols = 40, 16             # 6 * 40 = 240 bits per symbol: a multiple of 16, but ...
    tx.framebytes = raw.framebytes((6, (6, 4)), 40 * 16) - 1
    tx.modulate(b"\x00" * tx.framebytes)           # ... 15 columns of 3: still a permutation, accepted
    tx.ncarriers = 44                              # 264 bits: 16.5 columns -> not a permutation
    tx.framebytes = raw.framebytes((6, (6, 4)), 44 * 16) - 1
    with pytest.raises(_lib.OfdmTxError):
        tx.modulate(b"\x00" * tx.framebytes)


# ------------------------------------------------------------------------------ known answers derived by hand

def test_qam_gray_levels_by_hand():
    """The recursion of rawofdm/src/lib/qam/raw_qam.cc:63-74 (pt = bit * flip + 2 pt; flip *= -bit, with
    bit in {-1, +1}, first bit first) worked out on paper for every bit pattern:
      1 bit : 0 -> -1, 1 -> +1
      2 bits: 00 -> -3 (pt -1, flip +1; then -1 - 2), 01 -> -1, 11 -> +1, 10 -> +3  (Gray along the axis)
      3 bits: 000 -> -7, 001 -> -5, 011 -> -3, 010 -> -1, 110 -> +1, 111 -> +3, 101 -> +5, 100 -> +7
    and the scale of :39-54 with the powers of :120-147: BPSK sqrt(1), QPSK sqrt(1/2) per axis,
    16-QAM sqrt(0.5 * 2 / 10), 64-QAM sqrt(0.5 * 4 / 84).  An oracle that misread the recursion or the
    bit order fails here, whatever the GPU does."""
    by_hand = {1: {"0": -1, "1": 1},
               2: {"00": -3, "01": -1, "11": 1, "10": 3},
               3: {"000": -7, "001": -5, "011": -3, "010": -1, "110": 1, "111": 3, "101": 5, "100": 7}}
    for numbits, table in by_hand.items():
        for bits, level in table.items():
            got = ro.qam_axis([int(c) for c in bits], numbits, np.float32(1.0))
            assert got.tolist() == [float(level)], (bits, got)
    for nbits, scale2 in ((1, 1.0), (2, 0.5), (4, 0.5 * 2 / 10), (6, 0.5 * 4 / 84)):
        numbits, power = ro._QAM[nbits]
        assert abs(float(ro.qam_scale(numbits, power)) - np.sqrt(scale2)) < 1e-7
    # one whole 64-QAM value: items 1,0,0 | 0,1,1 -> (+7, -3) * sqrt(1/42)
    v = ro.qam_enc([1, 0, 0, 0, 1, 1], 6)
    assert np.allclose(v, (7 - 3j) * np.sqrt(1 / 42.0), rtol=1e-6)
    # unit mean power over the whole constellation (what the scale is for)
    allv = ro.qam_enc(np.array([[(i >> (5 - b)) & 1 for b in range(6)] for i in range(64)]).reshape(-1), 6)
    assert abs(np.mean(np.abs(allv) ** 2) - 1.0) < 1e-6


def test_bit_interleaver_map_by_hand():
    """rawofdm/src/lib/qam/raw_intrlv.cc:22-30 evaluated by hand for 8 carriers of 4 bits (cbps 32, s 2):
    i = 2 (k mod 16) + k div 16, then j = 2 (i div 2) + (i + 32 - i div 2) mod 2, i.e. the pairs (4m, 4m+1) stay
    and the pairs (4m+2, 4m+3) swap; and for 16 carriers of 2 bits (s 1: the first step only)."""
    want = [0, 3, 4, 7, 8, 11, 12, 15, 16, 19, 20, 23, 24, 27, 28, 31,
            1, 2, 5, 6, 9, 10, 13, 14, 17, 18, 21, 22, 25, 26, 29, 30]
    assert ro.intrlv_bit_map(8, 4).tolist() == want
    assert ro.intrlv_bit_map(16, 2).tolist() == list(range(0, 32, 2)) + list(range(1, 32, 2))
    assert ro.intrlv_bit_map(16, 1).tolist() == list(range(16))
    # the block scatters: out[map[j]] = in[j] (raw_intrlv.cc:69-73)
    items = np.arange(32) % 7
    out = ro.intrlv_bit(items, 8, 4, False)
    assert all(out[want[j]] == items[j] for j in range(32))


def test_encoder_taps_by_hand():
    """rawofdm/src/lib/qam/raw_conv.cc:46-51 with POLYS {121, 91} (spiral/defs.h:3-7): a single 1 bit
    followed by zeros reads the polynomials out, newest bit at the register's LSB: output pair n is
    (bit n of 121, bit n of 91) = 1111001b, 1011011b from the LSB up."""
    items, sr = ro.conv_enc(bytes([0x80, 0x00]))
    pairs = items.reshape(-1, 2)[:7].tolist()
    assert pairs == [[(121 >> n) & 1, (91 >> n) & 1] for n in range(7)]
    assert pairs == [[1, 1], [0, 1], [0, 0], [1, 1], [1, 1], [1, 0], [1, 1]]
    assert not items[14:].any() and sr == 0x8000
    # puncturing keeps the first np of every nc items (raw_conv.cc:150-159)
    assert ro.conv_punc(np.arange(12) % 5, 6, 4).tolist() == [0, 1, 2, 3, 1, 2, 3, 4]


# ------------------------------------------------------------------------------ caller surface

@pytest.mark.gpu
def test_mapper_block_alone(txctx):
    """raw.ofdm_mapper(carrier_map) (raw_ofdm.i:18-20) on its own: data to the carriers marked 1,
    alternating +1 / -1 on the carriers marked 2, bit-exact; the constructor's check of the map."""
    from ofdm_amd import raw
    rng = np.random.default_rng(12)
    p = ro.OfdmParams(ro.Options(fft_length=128, occupied_tones=100, cp_length=9))
    m = raw.ofdm_mapper(p.padded_carriers, ctx=txctx)
    assert (m.fft_length, m.data_tones) == (128, p.data_tones)
    data = (rng.standard_normal((11, p.data_tones)) + 1j * rng.standard_normal((11, p.data_tones))).astype(np.complex64)
    got = m.work(data).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), ro.mapper(p.padded_carriers, data).view(np.uint64))
    en = np.array([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1], np.uint8)
    got = m.work(data, en).cpu().numpy()
    assert np.array_equal(got, ro.mapper(p.padded_carriers, data, en))
    with pytest.raises(RuntimeError):                      # raw_ofdm_mapper.cc:37-38
        raw.ofdm_mapper([0, 1, 3, 1])


def test_make_data_is_the_reference_generator():
    """rawofdm/examples/ofdm_rxtx.py:101-121: unit QPSK from random.seed(78532); Python 2's randint
    is int(random() * 4) on that stream."""
    import cmath
    import math
    import random
    from ofdm_amd import ofdm_rxtx
    src = ofdm_rxtx.make_data(48, 5)
    random.seed(78532)
    want = np.array([cmath.exp(1j * (math.pi / 4 * (2 * int(random.random() * 4) + 1))) for _ in range(48 * 5)],
                    dtype=np.complex64)
    assert src.vlen == 48
    assert np.array_equal(src.take(5).numpy(), want)
    assert np.array_equal(src.take(7).numpy(), np.concatenate([want, want])[:7 * 48])      # repeat=True
    assert np.allclose(np.abs(want), 1.0) and len(set(np.round(want, 3).tolist())) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [False, True], ids=["allocations", "ring"])
@pytest.mark.parametrize("kw,burst,numpkts", [(dict(), 1, 5), (dict(fft_length=256, occupied_tones=208, cp_length=64, size=7), 3, 10)])
def test_tx_driven_like_sender_thread(txctx, tmp_path, monkeypatch, kw, burst, numpkts, ring):
    """rawofdm/examples/ofdm_tx.py:37-52 + ofdm_rxtx.py:194-215: make_data -> TX, sender_thread sends
    bursts of frames, then end of stream; the file holds the frames' samples (times --amp).
    ring: every batch counts as large, i.e. goes into the slots of the output ring, is scaled in place
    and reaches the sink one batch late."""
    from ofdm_amd import batch, gr, ofdm_rxtx
    if ring:
        monkeypatch.setattr(batch.StreamOutput, "RING_MIN_BYTES", 1)
    opt = ro.Options(**kw)
    opt.pause, opt.numpkts, opt.burst = 0.0, numpkts, burst
    tb = gr.top_block()
    tx = ofdm_rxtx.TX(opt, ctx=txctx, batch_frames=4)      # small batches: several kernel launches per run
    data = ofdm_rxtx.make_data(tx.params.data_tones, opt.size, None)
    tb.connect(data, tx)
    out = str(tmp_path / "tx.dat")
    tb.connect(tx, gr.multiply_const_cc(0.5), gr.file_sink(gr.sizeof_gr_complex, out))
    tb.start()
    sent = ofdm_rxtx.sender_thread(tx, opt).run()
    tb.wait()
    assert sent >= numpkts and sent % burst == 0
    assert (tx._output.ring is not None) == ring and (not ring or tx._output.ring.slots == 2)
    p = ro.OfdmParams(opt)
    one = data._data.reshape(opt.size, p.data_tones)
    ref = ro.modulate(p, np.tile(one, (sent, 1)), ro.frame_flags(sent, opt.size)) * 0.5
    got = np.fromfile(out, np.complex64)
    err = rel_l2(got, ref, p.fft_length + p.cp_length)
    assert err.max() <= REL_L2_TOL, err.max()
    # qa_ofdm.py:65-67: unit transmit power
    assert abs(np.mean(np.abs(got / 0.5) ** 2) * (p.fft_length / float(p.occupied_tones - 1)) - 1.0) < 0.2
