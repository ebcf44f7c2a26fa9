"""The C-ABI library loads on a machine without a GPU, exports every symbol include/ofdmtx.h
declares, and its host-only helpers agree with the oracle.  No kernels are launched here."""
import os
import re

import numpy as np
import pytest

from oracle import ftw_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ofdmtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdmtx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from ofdm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_functions()
    assert len(names) >= 15
    L = lib.load()
    for n in names:
        assert hasattr(L, n), "libofdmtx.so does not export %s" % n
    assert sorted(lib.PROTOTYPES) == names


def test_packet_info_matches_get_info(lib):
    rng = np.random.default_rng(9)
    for regime in range(1, 9):
        for n in [0, 1, 10, 1500, 3140] + rng.integers(0, 3141, 20).tolist():
            for T in (4.0, 8.0, 160.0):
                pi = lib.packet_info(regime, n, T)
                gi = o.get_info(b"\0" * n, str(regime), T)
                assert (pi.n_cbps, pi.n_bpsc, pi.n_dbps, pi.n_sym, pi.n_data, pi.n_pad, pi.txtime, pi.length) == \
                    (gi["N_cbps"], gi["N_bpsc"], gi["N_dbps"], gi["N_sym"], gi["N_data"], gi["N_pad"], gi["txtime"],
                     gi["packet_len"])
                assert pi.plcp_len == 5 + pi.length + o.tail_and_pad_bytes(pi.n_pad)
                assert pi.msg_len == 6 + pi.n_sym * pi.n_cbps // 8
                assert pi.n_rows == 4 + 1 + pi.n_sym + 13


def test_error_codes(lib):
    with pytest.raises(ValueError):
        lib.packet_info(1, 3141)
    with pytest.raises(lib.OfdmTxError) as e:
        lib.packet_info(9, 10)
    assert e.value.code == lib.EREGIME
    assert lib.load().ofdmtx_strerror(lib.ELENGTH) == b"MPDU-length must be in [0, 3168]"


def test_host_crc_matches_oracle(lib):
    rng = np.random.default_rng(1)
    for n in (0, 1, 24, 1524):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert lib.crc32(d) == o.ftw_crc32(d)


def test_update_crc32_is_the_uninverted_register(lib):
    """ftw/ftw_crc32.cc:33-64: bit-reversed CRC register from the all-ones start, the `crc` argument
    ignored; ftw_crc32 = that ^ 0xFFFFFFFF (:72-76)."""
    import zlib
    from ofdm_amd import ftw
    rng = np.random.default_rng(2)
    for n in (0, 1, 5, 100, 1524):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        reg = (zlib.crc32(d) ^ 0xFFFFFFFF) & 0xFFFFFFFF              # zlib inverts at the end; undo it
        want = int("{:032b}".format(reg)[::-1], 2)
        for seed in (0, 0xFFFFFFFF, 12345):
            assert ftw.ftw_update_crc32(seed, d) == want
        assert ftw.ftw_crc32(d) == want ^ 0xFFFFFFFF
