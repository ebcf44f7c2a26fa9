"""A minimal IEEE 802.11a OFDM receiver for noiseless frames, written from the standard (clause 17 of
IEEE 802.11-2007), NOT from the reference transmitter: FFT, carrier extraction, hard de-mapping by
the standard's Gray tables, de-interleaving (17.3.5.6), de-puncturing (17.3.5.5), Viterbi decoding of
the (133, 171) code, descrambling with the seed recovered from the SERVICE bits (17.3.5.4), SIGNAL
parsing (17.3.4) and the FCS check.

Test helper: it shows that what the transmitter (oracle or GPU) emits is a frame a standard receiver
decodes -- for every rate, which the Annex G vector alone (36 Mbit/s) cannot show.
"""
import zlib

import numpy as np

RATES = {0b1101: (1, "1/2"), 0b1111: (1, "3/4"), 0b0101: (2, "1/2"), 0b0111: (2, "3/4"),
         0b1001: (4, "1/2"), 0b1011: (4, "3/4"), 0b0001: (6, "2/3"), 0b0011: (6, "3/4")}   # Table 17-6: R1-R4 -> N_bpsc, rate
K_MOD = {1: 1.0, 2: 1 / np.sqrt(2), 4: 1 / np.sqrt(10), 6: 1 / np.sqrt(42)}             # Table 17-7
DATA_FREQS = [f for f in range(-26, 27) if f not in (-21, -7, 0, 7, 21)]                # 17.3.5.9


def carriers(row, gain):
    """One 80-sample symbol -> the 48 data carrier values, in increasing frequency order."""
    x = np.asarray(row, dtype=np.complex128)[16:]
    # the transmitter's rows are (gain / 8) * sum_k X[k] e^{+j 2 pi k n / 64}
    spec = np.fft.fft(x) / (8.0 * gain)
    return spec[[f % 64 for f in DATA_FREQS]]


def demap(vals, n_bpsc):
    """Hard decisions, Gray tables 17-8 .. 17-11: first half of the bits from I, second half from Q."""
    if n_bpsc == 1:
        return (vals.real > 0).astype(np.uint8)
    half = n_bpsc // 2
    out = np.zeros((len(vals), n_bpsc), dtype=np.uint8)
    for axis, comp in enumerate((vals.real, vals.imag)):
        lv = np.clip(np.round((comp / K_MOD[n_bpsc] + (2 ** half - 1)) / 2), 0, 2 ** half - 1).astype(int)  # 0 .. 2^half-1
        gray = lv ^ (lv >> 1)                                     # level index -> Gray code, MSB first
        for b in range(half):
            out[:, axis * half + b] = (gray >> (half - 1 - b)) & 1
    return out.reshape(-1)


def deinterleave(bits, n_bpsc):
    """Inverse of the two permutations of 17.3.5.6 for one symbol."""
    n = 48 * n_bpsc
    s = max(n_bpsc // 2, 1)
    k = np.arange(n)
    i = (n // 16) * (k % 16) + k // 16
    j = s * (i // s) + (i + n - (16 * i) // n) % s
    out = np.empty(n, dtype=bits.dtype)
    out[k] = bits[j]                                              # transmitter put bit k at position j
    return out


def depuncture(bits, rate):
    """17.3.5.5 (Figure 17-9): re-insert erasures (value 2) where bits were stolen."""
    if rate == "1/2":
        return bits
    period, kept = (6, [0, 1, 2, 5]) if rate == "3/4" else (4, [0, 1, 2])
    n_groups = len(bits) // len(kept)
    out = np.full((n_groups, period), 2, dtype=np.uint8)
    out[:, kept] = bits[:n_groups * len(kept)].reshape(n_groups, len(kept))
    return out.reshape(-1)


def viterbi(coded):
    """Hard-decision Viterbi decoder for g0 = 133, g1 = 171 (octal), K = 7, start state 0.
    coded: A0 B0 A1 B1 .. with 2 = erasure.  State = the previous six input bits, the most recent
    in bit 5; the encoder register is (input << 6) | state, so that g0 = 133 octal taps the input
    and the delays 2, 3, 5, 6 and g1 = 171 octal the input and the delays 1, 2, 3, 6."""
    coded = np.asarray(coded, dtype=np.int64).reshape(-1, 2)
    tgt = np.arange(64)
    bit = tgt >> 5                                               # the input bit that leads INTO a state
    pred = [((tgt & 31) << 1) | lo for lo in (0, 1)]             # its two possible predecessors
    par = lambda v: np.array([bin(int(x)).count("1") & 1 for x in v])
    exp_a = [par(((bit << 6) | p) & 0o133) for p in pred]
    exp_b = [par(((bit << 6) | p) & 0o171) for p in pred]
    metric = np.full(64, 1 << 30, dtype=np.int64)
    metric[0] = 0
    choice = np.zeros((len(coded), 64), dtype=np.uint8)
    for t, (a, b) in enumerate(coded):
        cost = []
        for lo in (0, 1):
            c = metric[pred[lo]].copy()
            if a != 2:
                c += exp_a[lo] != a
            if b != 2:
                c += exp_b[lo] != b
            cost.append(c)
        choice[t] = cost[1] < cost[0]
        metric = np.minimum(cost[0], cost[1])
    state = int(np.argmin(metric))
    bits = np.zeros(len(coded), dtype=np.uint8)
    for t in range(len(coded) - 1, -1, -1):
        bits[t] = state >> 5
        state = ((state & 31) << 1) | int(choice[t, state])
    return bits


def descramble(bits):
    """17.3.5.4: S(x) = x^7 + x^4 + 1, i.e. the sequence obeys s[n] = s[n-4] ^ s[n-7]; the first
    seven SERVICE bits are zero before scrambling, so the received ones ARE s[0..6]."""
    seq = list(bits[:7])
    for n in range(7, len(bits)):
        seq.append(seq[n - 4] ^ seq[n - 7])
    return np.asarray(bits, dtype=np.uint8) ^ np.asarray(seq, dtype=np.uint8)


def receive(iq, gain=0.3, swap_iq=False):
    """iq: one PPDU (320 preamble samples, SIGNAL, DATA symbols; anything behind them is ignored).
    Returns dict(rate_bits, n_bpsc, rate, length, psdu, fcs_ok, service, n_sym)."""
    iq = np.asarray(iq, dtype=np.complex128)
    if swap_iq:
        iq = iq.imag + 1j * iq.real
    rows = iq[320:].reshape(-1, 80)
    # SIGNAL: BPSK, rate 1/2, 24 bits (17.3.4)
    sig = viterbi(deinterleave(demap(carriers(rows[0], gain), 1), 1))
    rate_bits = int("".join(map(str, sig[0:4])), 2)
    length = sum(int(sig[5 + i]) << i for i in range(12))
    parity_ok = (int(sig[:17].sum()) + int(sig[17])) % 2 == 0
    assert sig[4] == 0 and parity_ok and not sig[18:24].any(), "SIGNAL field does not parse"
    n_bpsc, rate = RATES[rate_bits]
    n_cbps = 48 * n_bpsc
    n_dbps = n_cbps * {"1/2": 1, "2/3": 2, "3/4": 3}[rate] // {"1/2": 2, "2/3": 3, "3/4": 4}[rate]
    n_sym = -(-(16 + 8 * length + 6) // n_dbps)
    coded = np.concatenate([depuncture(deinterleave(demap(carriers(rows[1 + s], gain), n_bpsc), n_bpsc), rate)
                            for s in range(n_sym)])
    scrambled = viterbi(coded)
    data = descramble(scrambled)
    psdu_bits = data[16:16 + 8 * length].reshape(-1, 8)
    psdu = bytes(int(sum(int(b) << i for i, b in enumerate(byte))) for byte in psdu_bits)     # LSB first (17.3.5.1)
    fcs_ok = len(psdu) >= 4 and (zlib.crc32(psdu[:-4]) & 0xFFFFFFFF) == int.from_bytes(psdu[-4:], "little")
    return {"rate_bits": rate_bits, "n_bpsc": n_bpsc, "rate": rate, "length": length, "psdu": psdu,
            "fcs_ok": fcs_ok, "service": data[:16], "n_sym": n_sym}
