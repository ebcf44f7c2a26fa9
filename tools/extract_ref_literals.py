#!/usr/bin/env python3
"""Extract the numeric literal tables of the reference as DATA (text parsing only).

The reference's files are read as text; nothing from them is imported, compiled
or executed.  The numbers found are written to tests/golden/ftw_tables.json,
which is the fixture the oracle loads and against which the formula-generated
device tables are checked.

Tables extracted
  ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py:346-349  scrambling_seq (127 bits)
  ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py:497-506  gray_const_{bpsk,qpsk,qam16,qam64}
  ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py:508      fft_preamble (320 complex)
  ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py:510      gap_sample (80 complex zeros)
  ftw_80211_ofdm_tx/src/ftw/ftw_ofdm_pilot_cc.cc:96-104       polarity (127 values)
  ftw_80211_ofdm_tx/src/matlab/802.11-2007-Annex-G.8.txt      881 time-domain samples

Run:  python tools/extract_ref_literals.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

_NUM = r"[+-]?\s*\d+(?:\.\d*)?"


def _parse_complex(tok):
    """'0.3680 + 0.3680j' | '0.7360' | '0 + 0.7360j' | '-1+0j' -> (re, im)."""
    t = tok.replace(" ", "")
    m = re.fullmatch(r"([+-]?\d+(?:\.\d*)?)([+-]\d+(?:\.\d*)?)j", t)
    if m:
        return float(m.group(1)), float(m.group(2))
    m = re.fullmatch(r"([+-]?\d+(?:\.\d*)?)j", t)
    if m:
        return 0.0, float(m.group(1))
    m = re.fullmatch(r"([+-]?\d+(?:\.\d*)?)", t)
    if m:
        return float(m.group(1)), 0.0
    raise ValueError("cannot parse complex literal %r" % tok)


def _tuple_after(text, name):
    """Text between the parentheses of `name = ( ... )`."""
    m = re.search(r"^%s\s*=\s*\(" % re.escape(name), text, re.M)
    if not m:
        raise KeyError(name)
    start = m.end()
    end = text.index(")", start)
    return text[start:end]


def complex_table(text, name):
    body = _tuple_after(text, name).replace("\n", " ")
    toks = [t for t in (s.strip() for s in body.split(",")) if t]
    return [list(_parse_complex(t)) for t in toks]


def int_list_after(text, anchor, count):
    i = text.index(anchor)
    j = text.index("[", i) if "[" in text[i:i + 200] else text.index("{", i)
    vals = re.findall(r"-?\d+", text[j:])
    return [int(v) for v in vals[:count]]


def main():
    pu = open(os.path.join(REF, "ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py")).read()
    pc = open(os.path.join(REF, "ftw_80211_ofdm_tx/src/ftw/ftw_ofdm_pilot_cc.cc")).read()
    ag = open(os.path.join(REF, "ftw_80211_ofdm_tx/src/matlab/802.11-2007-Annex-G.8.txt")).read()

    out = {}
    out["scrambling_seq"] = int_list_after(pu, "scrambling_seq = [", 127)
    for name in ("gray_const_bpsk", "gray_const_qpsk", "gray_const_qam16",
                 "gray_const_qam64", "fft_preamble", "gap_sample"):
        out[name] = complex_table(pu, name)
    # polarity[127] = { 1, 1, ... } in the pilot block
    i = pc.index("polarity[127]")
    j = pc.index("{", i)
    k = pc.index("}", j)
    out["pilot_polarity"] = [int(v) for v in re.findall(r"-?\d+", pc[j:k])]

    # Annex G.8: triples (index, re, im), one number per line
    nums = ag.split()
    assert len(nums) % 3 == 0
    rows = []
    for r in range(len(nums) // 3):
        idx = int(nums[3 * r])
        assert idx == r, (idx, r)
        rows.append([float(nums[3 * r + 1]), float(nums[3 * r + 2])])
    out["annex_g8_time"] = rows

    assert len(out["scrambling_seq"]) == 127
    assert len(out["pilot_polarity"]) == 127
    assert [len(out[n]) for n in ("gray_const_bpsk", "gray_const_qpsk", "gray_const_qam16",
                                  "gray_const_qam64", "fft_preamble", "gap_sample")] == [2, 4, 16, 64, 320, 80]
    assert len(rows) == 881

    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "ftw_tables.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", os.path.normpath(path), {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
