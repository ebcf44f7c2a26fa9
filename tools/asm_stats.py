#!/usr/bin/env python3
"""Static instruction mix of one kernel in build/ofdmtx.s (after `make -C ofdm_amd/csrc asm`).
usage: tools/asm_stats.py <mangled-name-substring>"""
import collections
import re
import sys

name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "build/ofdmtx.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
c = collections.Counter()
for l in lines[start:end]:
    t = l.strip().split(" ")[0] if l.strip() else ""
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    if t.startswith("v_pk_"):
        c["valu_packed"] += 1
    elif t.startswith("v_"):
        c["valu"] += 1
    elif t.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif t.startswith("s_cbranch") or t.startswith("s_branch"):
        c["branch"] += 1
    elif t.startswith("s_"):
        c["salu"] += 1
    elif t.startswith("ds_"):
        c["lds"] += 1
    elif t.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["vmem"] += 1
    else:
        c["other"] += 1
print(name, dict(c), "total", sum(c.values()))
