"""The structure the kernels' interleaving-by-multiplication rests on, checked on the CPU against the
oracle's interleaver (ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py:231-277) and puncturing
(:310-328): which row and column of the 16-column block feeds which constellation-index bit of which
carrier, where those bits lie in the encoder's A and B streams, and that the multiplier constants of
`namespace spread` (ofdm_amd/csrc/ofdmtx.hip), re-derived here from the same rules, put every kept
bit where it belongs for every value of a field.  No GPU: this pins the derivation, the GPU parity
tests pin the kernels."""
import numpy as np
import pytest

from oracle import ftw_oracle as fo

CASES = [(2, 0), (2, 1), (4, 0), (4, 1), (6, 1)]      # (N_bpsc, puncturing: 0 = rate 1/2, 1 = rate 3/4): regimes 3, 4, 5, 6, 8


def coded_source(n_bpsc, punct):
    """For every interleaved bit j of a DATA symbol: (stream, n) = bit n of the encoder's A (0) or B (1)
    stream of that symbol, from the oracle's permutation and the reference's puncturing rule."""
    n_cbps = 48 * n_bpsc
    dest = fo.interleave_perm(n_cbps, n_bpsc)                      # punctured position k -> interleaved position
    src = np.empty(n_cbps, dtype=np.int64)
    src[dest] = np.arange(n_cbps)
    out = []
    for j in range(n_cbps):
        q = int(src[j])
        p = q if punct == 0 else 6 * (q // 4) + (5 if q % 4 == 3 else q % 4)     # keep p % 6 in {0,1,2,5}
        out.append((p & 1, p >> 1))
    return out


@pytest.mark.parametrize("n_bpsc,punct", CASES)
def test_block_structure(n_bpsc, punct):
    """Row r = N_bpsc g + s h + i0 of the block, column c, feeds index bit s h + (i0 - c) mod s of carrier
    3c + g; the column's bit is bit U r + phi(c) of stream sigma(c) (U = 8 or 12 stream bits per row)."""
    s = max(n_bpsc // 2, 1)
    unit = 8 if punct == 0 else 12
    src = coded_source(n_bpsc, punct)
    seen = set()
    for g in range(3):
        for rho in range(n_bpsc):
            r, h, i0 = n_bpsc * g + rho, rho // s, rho % s
            for c in range(16):
                t = s * h + (i0 - c) % s
                m = 3 * c + g
                sigma = c & 1
                if punct == 0:
                    phi = c // 2
                else:
                    phi = 3 * (c // 4) + (0 if c % 4 < 2 else 1 if c % 4 == 2 else 2)
                assert src[n_bpsc * m + t] == (sigma, unit * r + phi)
                seen.add(n_bpsc * m + t)
    assert len(seen) == 48 * n_bpsc


def constants(n_bpsc, punct, sigma, q, i0):
    """Multiplier and mask of one (stream, half, row class), and the field's kept bits (value bit 0 = last stream bit)."""
    s = max(n_bpsc // 2, 1)
    kept = [0, 1, 2, 3] if punct == 0 else ([0, 1, 3, 4] if sigma == 0 else [0, 2, 3, 5])
    mult = mask = 0
    pos = []
    for b in range(4):
        c = 8 * q + 2 * (3 - b) + sigma
        p = s - 1 - (i0 - c) % s
        pos.append(8 * b + p)
        mult |= 1 << (8 * b + p - kept[b])
        mask |= 1 << (8 * b + p)
    return mult, mask, kept, pos


@pytest.mark.parametrize("n_bpsc,punct", CASES)
def test_multiplier_constants(n_bpsc, punct):
    s = max(n_bpsc // 2, 1)
    masks = set()
    for sigma in range(2):
        width = 4 if punct == 0 else 5 + sigma
        for q in range(2):
            for i0 in range(s):
                mult, mask, kept, pos = constants(n_bpsc, punct, sigma, q, i0)
                assert mult < 1 << 24                               # a 24-bit multiply
                masks.add(mask)
                for f in range(1 << width):
                    if any((f >> k) & 1 for k in range(width) if k not in kept):
                        continue                                    # dropped bits are cleared beforehand
                    want = sum(((f >> kept[b]) & 1) << pos[b] for b in range(4))
                    assert (f * mult) & 0xFFFFFFFF & mask == want
    assert len(masks) == s                                          # one mask per rotation of the rows


@pytest.mark.parametrize("n_bpsc,punct", CASES)
def test_symbol_by_multiplication_equals_oracle(n_bpsc, punct):
    """A whole symbol: carrier indices formed from random A / B streams by fields, multipliers and masks equal
    the ones read off the oracle's puncturing + interleaving."""
    s = max(n_bpsc // 2, 1)
    unit = 8 if punct == 0 else 12
    n_dbps = 3 * n_bpsc * unit
    rng = np.random.default_rng(100 * n_bpsc + punct)
    src = coded_source(n_bpsc, punct)
    for _ in range(5):
        streams = rng.integers(0, 2, (2, n_dbps))
        want = [sum(int(streams[src[n_bpsc * m + t]]) << (n_bpsc - 1 - t) for t in range(n_bpsc)) for m in range(48)]
        got = [None] * 48
        for g in range(3):
            for sigma in range(2):
                width = 4 if punct == 0 else 5 + sigma
                for q in range(2):
                    acc = [0, 0]
                    for rho in range(n_bpsc):
                        r, h, i0 = n_bpsc * g + rho, rho // s, rho % s
                        mult, mask, kept, _ = constants(n_bpsc, punct, sigma, q, i0)
                        n0 = unit * r + (unit // 2) * q
                        f = 0
                        for e in range(width):
                            if width - 1 - e in kept:
                                f |= int(streams[sigma, n0 + e]) << (width - 1 - e)
                        acc[h] |= (f * mult) & mask
                    word = (acc[0] << s) | acc[1]
                    for b in range(4):
                        c = 8 * q + 2 * (3 - b) + sigma
                        got[3 * c + g] = (word >> (8 * b)) & 0xFF
        assert got == want
