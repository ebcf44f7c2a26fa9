"""CPU ORACLE, third restatement (test infrastructure, not product code): the bit-level chain of
the FTW transmitter one bit at a time in plain Python, keeping the reference's CONTROL FLOW --
scatter loops for the permutations, the sliding-offset loops of the puncturer, a shift register
for the encoder -- where the numpy oracle (oracle/ftw_oracle.py) uses closed forms (index masks,
shifted XORs, inverse permutations).  Agreement of the two pins the closed forms; its run time
also shows the order of magnitude of a per-bit Python transmitter, which bench.py reports beside
the C port's as cpu_baseline.config1_py_refshaped (one packet, outside every timed region).

Only tests/ and the cpu_baseline leg of bench.py may import this module.  Citations: pu = ftw_80211_ofdm_tx/src/examples/
ftw_packet_utils.py of the reference.
"""
import zlib

RATE_BITS = {"1": 0x0D, "2": 0x0F, "3": 0x05, "4": 0x07, "5": 0x09, "6": 0x0B, "7": 0x01, "8": 0x03}   # pu:129-144
GEOMETRY = {"1": (48, 1, 24), "2": (48, 1, 36), "3": (96, 2, 48), "4": (96, 2, 72),
            "5": (192, 4, 96), "6": (192, 4, 144), "7": (288, 6, 192), "8": (288, 6, 216)}             # pu:39-81

MAC_HEADER_TAIL = bytes.fromhex("006008cd37a6" "0020d6013cf1" "006008ad3baf" "0000")                  # pu:30-33


def to_bits(data):
    return [(byte >> (7 - i)) & 1 for byte in data for i in range(8)]                                  # pu:406-417


def to_bytes(bits):
    assert len(bits) % 8 == 0
    out = bytearray()
    for i in range(0, len(bits), 8):
        v = 0
        for b in bits[i:i + 8]:
            v = (v << 1) | b
        out.append(v)
    return bytes(out)


def lfsr_sequence():
    """127 scrambler bits for the state 1011101 (x7..x1) the reference hard-wires (pu:346-349)."""
    state = [1, 0, 1, 1, 1, 0, 1]                        # x7 .. x1
    seq = []
    for _ in range(127):
        fb = state[0] ^ state[3]                         # x7 xor x4
        seq.append(fb)
        state = state[1:] + [fb]
    return seq


def plcp(payload, regime, symboltime):
    """pu:25-205: PLCP byte string and Length."""
    n_cbps, n_bpsc, n_dbps = GEOMETRY[regime]
    body_len = 24 + len(payload)
    n_bits = 16 + 32 + 6 + 8 * body_len
    n_sym = -(-n_bits // n_dbps)
    txtime = int(5 * symboltime + symboltime * n_sym) + 2
    mpdu = bytes([0x08, 0x00, (txtime >> 8) & 0xFF, txtime & 0xFF]) + MAC_HEADER_TAIL + bytes(payload)
    n_pad = n_sym * n_dbps - n_bits
    length = body_len + 4
    # LENGTH with its least significant bit first (pu:147-150), even parity over RATE and LENGTH (pu:153-157)
    rev = 0
    for i in range(12):
        rev |= ((length >> i) & 1) << (11 - i)
    ones = bin(rev).count("1") + bin(RATE_BITS[regime]).count("1")
    field = (RATE_BITS[regime] << 20) | (rev << 7) | ((ones % 2) << 6)
    signal = bytes([(field >> 16) & 0xFF, (field >> 8) & 0xFF, field & 0xFF])
    # every MPDU byte with its bits in transmit order (pu:208-218)
    flipped = bytearray()
    for byte in mpdu:
        r = 0
        for i in range(8):
            r |= ((byte >> i) & 1) << (7 - i)
        flipped.append(r)
    # FCS: reflected CRC-32 with its 32 bits reversed (ftw/ftw_crc32.cc:52-63,75), big-endian (pu:372)
    crc = zlib.crc32(mpdu) & 0xFFFFFFFF
    fcs = 0
    for i in range(32):
        fcs |= ((crc >> i) & 1) << (31 - i)
    n_tail = (n_pad + 6) // 8 + (0 if (n_pad + 6) % 8 == 0 else 1)                                     # pu:160-171
    return signal + bytes(2) + bytes(flipped) + fcs.to_bytes(4, "big") + bytes(n_tail), length, n_sym


def scramble(pkt, length):
    """pu:344-367."""
    bits = to_bits(pkt)
    seq = lfsr_sequence()
    out = bits[:24]
    for k in range(24, len(bits)):
        out.append(bits[k] ^ seq[(k - 24) % 127])
    z = 8 * length
    for i in range(23 + z + 17, 23 + z + 22 + 1):
        out[i] = 0
    return to_bytes(out)


def encode(pkt, regime, n_sym):
    """pu:280-341: shift-register encoder, then the puncturer's sliding-offset loops."""
    n_cbps = GEOMETRY[regime][0]
    wanted = n_sym * n_cbps
    coded = []
    reg = 0
    for bit in to_bits(pkt):
        reg = (reg >> 1) | (bit << 6)                                                                  # pu:292-297
        coded.append(bin(reg & 0x5B).count("1") & 1)                                                   # g0 = 133 octal
        coded.append(bin(reg & 0x79).count("1") & 1)                                                   # g1 = 171 octal
    if regime in ("1", "3", "5"):
        return to_bytes(coded[:48 + wanted])
    signal, data = coded[:48], coded[48:]
    kept = data[0:3]
    off = 0
    if regime == "7":                                                                                  # pu:330-341
        while off + 4 < len(data) - 1:
            kept += data[4 + off:7 + off]
            off += 4
    else:                                                                                              # pu:316-328
        while off + 9 < len(data) - 1:
            kept += data[5 + off:9 + off]
            off += 6
        kept.append(data[5 + off])
    return to_bytes(signal + kept[:wanted])


def interleave(coded, regime):
    """pu:222-277: both permutations as scatter loops, SIGNAL symbol first step only."""
    n_cbps, n_bpsc, _ = GEOMETRY[regime]
    src = to_bits(coded)
    n_sym = (len(src) - 48) // n_cbps
    half = [0] * len(src)
    for k in range(48):
        half[3 * (k % 16) + k // 16] = src[k]
    for sym in range(n_sym):
        base = 48 + sym * n_cbps
        for k in range(n_cbps):
            half[base + (n_cbps // 16) * (k % 16) + k // 16] = src[base + k]
    out = half[:48] + [0] * (len(src) - 48)
    s = max(n_bpsc // 2, 1)
    for sym in range(n_sym):
        base = 48 + sym * n_cbps
        for i in range(n_cbps):
            j = s * (i // s) + (i + n_cbps - (16 * i) // n_cbps) % s
            out[base + j] = half[base + i]
    return to_bytes(out)


def encode_bits(payload, regime, symboltime=4.0):
    pkt, length, n_sym = plcp(payload, regime, symboltime)
    scr = scramble(pkt, length)
    cod = encode(scr, regime, n_sym)
    return {"plcp": pkt, "Length": length, "N_sym": n_sym, "scrambled": scr, "coded": cod,
            "interleaved": interleave(cod, regime)}
