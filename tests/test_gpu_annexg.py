"""The reference's one golden vector through the HIP kernels: the time-domain frame of IEEE
802.11-2007 Annex G.8 (ftw_80211_ofdm_tx/src/matlab/802.11-2007-Annex-G.8.txt, committed as data in
tests/golden/ftw_tables.json) against the stage operators ofdmtx_scramble -> ofdmtx_conv_encode ->
ofdmtx_interleave and the fused symbol-level kernel ofdmtx_modulate, all through the C ABI.

The frame (36 Mbit/s = regime 6, 100-byte PSDU) carries a MAC header with frame control 0x0402 and the
FCS printed in the standard, which is not the CRC-32 of the MPDU (matlab/encode.m:55-66,221-223: "force
the WRONG CRC32 for the reference frame").  With their ordinary plans ofdmtx_tx / ofdmtx_encode_bits build
the MAC header of examples/ftw_packet_utils.py:28-36 (frame control 0x0800) and a correct FCS themselves,
so the frame enters in two other ways:
  * at the PLCP byte string, which the first test assembles by hand from the field definitions, into the
    stage operators and ofdmtx_modulate;
  * as a raw PSDU (plan flag OFDMTX_F_RAW_PSDU: the caller's bytes are MPDU + FCS) into the batched
    kernels themselves -- ofdmtx_tx, the fused payload -> IQ kernel the headline is measured on, and
    ofdmtx_encode_bits with all its taps.
Either way the kernels downstream (scrambler, encoder + 3/4 puncturing, 16-QAM interleaver, mapper,
pilots, carrier map, inverse FFT, cyclic prefix, 1/8 scale, preamble) see exactly the bits of the
standard's example.  Tolerance: the table has 3 decimals (half an LSB in re
and im: 7.1e-4) and is time-windowed at the symbol boundaries, which are left out
(matlab/encode.m:421-446).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from ofdm_amd import _lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = bytes.fromhex(
    "4a6f792c2062726967687420737061726b206f6620646976696e6974792c0a4461756768746572206f6620456c797369756d2c0a"
    "466972652d696e73697265642077652074726561")
# matlab/encode.m:59-65; duration 0x002e = TxTime of :150
MPDU = bytes.fromhex("0402" "002e" "006008cd37a6" "0020d6013cf1" "006008ad3baf" "0000") + BODY
# matlab/encode.m:223, transmit order
FCS_BITS = "01011011111010101001100110110111"


def annex_g_plcp():
    """SIGNAL (RATE 1011 = 36 Mbit/s, reserved 0, LENGTH 100 LSB first, even parity, six tail zeros),
    SERVICE (16 zeros), the PSDU with every byte sent LSB first, the printed FCS, tail + pad."""
    length = len(MPDU) + 4
    assert length == 100
    signal = "1011" + "0" + "{:012b}".format(length)[::-1]
    signal += str(signal.count("1") % 2) + "000000"
    psdu = "".join("{:08b}".format(b)[::-1] for b in MPDU) + FCS_BITS
    n_dbps = 144
    n_bits = 16 + 8 * length + 6
    n_sym = -(-n_bits // n_dbps)
    assert n_sym == 6
    n_pad = n_sym * n_dbps - n_bits
    bits = signal + "0" * 16 + psdu + "0" * (8 * ((n_pad + 6 + 7) // 8))     # pu:160-171 rounds tail+pad up to bytes
    assert len(bits) % 8 == 0
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)), length, n_sym


def stage(ctx, name, data, out_len, *args):
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(ctx.device)
    dst = torch.zeros(out_len, dtype=torch.uint8, device=ctx.device)
    st = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
    _lib.check(getattr(ctx._lib, name)(ctx._h, C.c_void_p(src.data_ptr()), len(data), *args,
                                       C.c_void_p(dst.data_ptr()), st))
    torch.cuda.synchronize()
    return dst.cpu().numpy().tobytes()


def test_annex_g8_frame_through_the_kernels(txctx):
    with open(os.path.join(ROOT, "tests", "golden", "ftw_tables.json")) as f:
        table = np.array([complex(r, i) for r, i in json.load(f)["annex_g8_time"]])
    assert len(table) == 881

    plcp, length, n_sym = annex_g_plcp()
    scr = stage(txctx, "ofdmtx_scramble", plcp, len(plcp), length)
    cod = stage(txctx, "ofdmtx_conv_encode", scr, 6 + n_sym * 24, 6, n_sym)
    itl = stage(txctx, "ofdmtx_interleave", cod, len(cod), 6)
    assert scr[:3] == plcp[:3]                                   # SIGNAL is not scrambled

    # one packet of the same geometry: 72-byte body -> 6 DATA symbols; amplitude 1, preamble, no gap,
    # pilot polarity from index 0 (the first packet of a stream)
    plan = txctx.plan_uniform(1, 6, len(BODY), pilot_mode=_lib.PILOT_PER_PACKET, pilot_start=0,
                              flags=_lib.F_PREAMBLE, amp=1.0)
    assert plan.layout()["n_sym"][0] == n_sym and plan.layout()["msg_len"][0] == len(itl)
    msg = torch.zeros(plan.msg_bytes, dtype=torch.uint8, device=txctx.device)
    msg[:len(itl)] = torch.frombuffer(bytearray(itl), dtype=torch.uint8).to(txctx.device)
    iq = plan.modulate(msg)
    torch.cuda.synchronize()
    got = iq.cpu().numpy()[:880].astype(np.complex128) / 8.0     # matlab/encode.m:421-446: frame = 8 x table
    d = np.abs(got - table[:880])
    boundary = np.zeros(880, bool)
    boundary[[0, 160, 320] + list(range(400, 880, 80))] = True
    assert d[~boundary].max() < 1.5e-3, "HIP output differs from the Annex G.8 table by %.3g" % d[~boundary].max()
    assert d[boundary].min() > 0.02                              # the table's windowed samples really differ


def annex_g_psdu():
    """The 100 bytes a raw-PSDU plan takes: the MPDU and, behind it, the printed FCS packed like every
    other PSDU byte (first transmitted bit = least significant)."""
    fcs = bytes(int(FCS_BITS[i:i + 8][::-1], 2) for i in range(0, 32, 8))
    return MPDU + fcs


def check_against_table(iq880):
    with open(os.path.join(ROOT, "tests", "golden", "ftw_tables.json")) as f:
        table = np.array([complex(r, i) for r, i in json.load(f)["annex_g8_time"]])
    d = np.abs(iq880.astype(np.complex128) / 8.0 - table[:880])  # matlab/encode.m:421-446: frame = 8 x table
    boundary = np.zeros(880, bool)
    boundary[[0, 160, 320] + list(range(400, 880, 80))] = True
    assert d[~boundary].max() < 1.5e-3, "HIP output differs from the Annex G.8 table by %.3g" % d[~boundary].max()
    assert d[boundary].min() > 0.02


@pytest.mark.parametrize("ragged", [False, True], ids=["uniform", "ragged"])
def test_annex_g8_frame_through_the_fused_kernel(txctx, ragged):
    """The PSDU through ofdmtx_tx (k_tx, payload input): SIGNAL field, scrambler words, encoder,
    puncturing and interleaving by multiplication, slot rows, transform -- one kernel, against the table.
    The ragged plan puts two more packets around it, so the frame is not the batch's first."""
    psdu = annex_g_psdu()
    flags = _lib.F_PREAMBLE | _lib.F_RAW_PSDU
    if ragged:
        other = bytes(range(40))
        plan = txctx.plan_ragged(np.array([3, 6, 8], np.uint8), np.array([len(other), len(psdu), len(other)], np.uint32),
                                 pilot_mode=_lib.PILOT_PER_PACKET, pilot_start=0, flags=flags, amp=1.0)
        data, which = other + psdu + other, 1
    else:
        plan = txctx.plan_uniform(1, 6, len(psdu), pilot_mode=_lib.PILOT_PER_PACKET, pilot_start=0, flags=flags, amp=1.0)
        data, which = psdu, 0
    lay = plan.layout()
    assert lay["n_sym"][which] == 6 and lay["plcp_len"][which] == len(annex_g_plcp()[0])
    pay = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(txctx.device)
    iq = plan.tx(pay)
    torch.cuda.synchronize()
    a = int(lay["sample_off"][which])
    check_against_table(iq.cpu().numpy()[a:a + 880])
    # and bit for bit what the two-kernel path makes of the same plan
    iq2 = plan.modulate(plan.encode_bits(pay))
    assert torch.equal(iq.view(torch.float32), iq2.view(torch.float32))


def test_annex_g8_bit_chain_taps_equal_the_oracle(txctx):
    """ofdmtx_encode_bits (k_encode_bits) on the PSDU: every tap bit for bit what the oracle's chain gives
    for the same MPDU and FCS (oracle.encode_bits_from_mpdu, itself pinned to the table by
    tests/test_oracle.py), and the PLCP tap equal to the byte string assembled by hand above."""
    from oracle import ftw_oracle as o
    psdu = annex_g_psdu()
    plan = txctx.plan_uniform(1, 6, len(psdu), pilot_mode=_lib.PILOT_PER_PACKET, flags=_lib.F_PREAMBLE | _lib.F_RAW_PSDU,
                              amp=1.0)
    taps = plan.encode_bits(torch.frombuffer(bytearray(psdu), dtype=torch.uint8).to(txctx.device), taps=True)
    torch.cuda.synchronize()
    ref = o.encode_bits_from_mpdu(MPDU, "6", o.bits_to_bytes(np.array([int(c) for c in FCS_BITS], np.uint8)))
    lay = plan.layout()
    plen, mlen = int(lay["plcp_len"][0]), int(lay["msg_len"][0])
    assert ref["plcp"] == annex_g_plcp()[0]
    for tap, key, n in (("plcp", "plcp", plen), ("scrambled", "scrambled", plen), ("coded", "coded", mlen),
                        ("interleaved", "interleaved", mlen)):
        assert taps[tap].cpu().numpy()[:n].tobytes() == ref[key], tap
