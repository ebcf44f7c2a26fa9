"""Device tables are generated from the PHY definitions (tools/gen_tables.py); they must equal the
literals the reference embeds (fixture tests/golden/ftw_tables.json, extracted as text)."""
import os
import re
import subprocess
import sys

import numpy as np

from oracle import ftw_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_tables as g  # noqa: E402


def test_scrambler_and_pilot_sequences_are_the_lfsr():
    t = o.tables()
    assert g.scrambling_seq() == t.scrambling_seq.tolist()
    assert g.pilot_polarity() == [int(v) for v in t.pilot_polarity]


def test_constellations_equal_reference_literals():
    t = o.tables()
    for nb, name in ((1, "bpsk"), (2, "qpsk"), (4, "qam16"), (6, "qam64")):
        assert np.array_equal(g.constellation(nb), t.constellation[name])


def test_preamble_equals_reference_literals():
    import json
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "ftw_tables.json")))["fft_preamble"]
    lit = np.array([complex(a, b) for a, b in lit])
    assert np.array_equal(g.preamble320(), lit)
    # 10 short symbols of 16 samples, then GI2 + 2 long symbols
    p = lit
    assert all(np.array_equal(p[:16], p[16 * i:16 * i + 16]) for i in range(10))
    assert np.array_equal(p[160:192], p[224:256]) and np.array_equal(p[192:256], p[256:320])


def _header_array(text, name):
    m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, text, re.S)
    return m.group(1)


def test_generated_header_is_current(tmp_path):
    out = tmp_path / "ofdm_tables.h"
    old = g.OUT
    try:
        g.OUT = str(out)
        g.main()
    finally:
        g.OUT = old
    assert out.read_text() == open(os.path.join(ROOT, "ofdm_amd", "csrc", "ofdm_tables.h")).read()


def test_header_values_match_fixture():
    text = open(os.path.join(ROOT, "ofdm_amd", "csrc", "ofdm_tables.h")).read()
    t = o.tables()
    cons = np.array([float(v.rstrip("f")) for v in re.findall(r"-?\d+\.\d+f", _header_array(text, "OFDMTX_CONST_DEFAULT"))])
    want = np.concatenate([t.constellation[m] for m in ("bpsk", "qpsk", "qam16", "qam64")])
    assert np.array_equal(cons[0::2] + 1j * cons[1::2], want)
    words = [int(v.rstrip("u"), 16) for v in re.findall(r"0x[0-9a-f]{8}u", _header_array(text, "OFDMTX_SCR_WORD"))]
    seq = t.scrambling_seq
    for ph in (0, 1, 8, 100, 126):
        assert words[ph] == int("".join(str(seq[(ph + i) % 127]) for i in range(32)), 2)
    neg = [int(v.rstrip("u"), 16) for v in re.findall(r"0x[0-9a-f]{8}u", _header_array(text, "OFDMTX_PILOT_NEG"))]
    for i in range(127):
        assert ((neg[i // 32] >> (i % 32)) & 1) == (1 if t.pilot_polarity[i] < 0 else 0)


def test_fixture_matches_reference_text_when_present(tmp_path):
    ref = "/root/reference"
    if not os.path.isdir(ref):
        import pytest
        pytest.skip("reference tree not present on this machine")
    env = dict(os.environ)
    script = os.path.join(ROOT, "tools", "extract_ref_literals.py")
    src = open(script).read().replace('OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")',
                                      'OUT = %r' % str(tmp_path))
    tmp_script = tmp_path / "extract.py"
    tmp_script.write_text(src)
    subprocess.check_call([sys.executable, str(tmp_script), ref], env=env, stdout=subprocess.DEVNULL)
    assert (tmp_path / "ftw_tables.json").read_text() == open(os.path.join(ROOT, "tests", "golden", "ftw_tables.json")).read()
