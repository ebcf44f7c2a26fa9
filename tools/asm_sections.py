#!/usr/bin/env python3
"""Per-section static instruction counts of one kernel, using the `; SEC name` markers that
-DOFDMTX_ASM_MARKERS puts into the assembly.
usage: tools/asm_sections.py <kernel-name-substring> [asm file]"""
import collections
import re
import sys

name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "build/ofdmtx_marked.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
sec = "head"
order = [sec]
cnt = {sec: collections.Counter()}
for l in lines[start:end]:
    m = re.search(r"; SEC (\w+)", l)
    if m:
        sec = "%s@%d" % (m.group(1), len(order))
        order.append(sec)
        cnt[sec] = collections.Counter()
        continue
    t = l.strip().split(" ")[0] if l.strip() else ""
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    k = ("pk" if t.startswith("v_pk_") else "v" if t.startswith("v_") else "wait" if t.startswith("s_waitcnt")
         else "br" if t.startswith(("s_cbranch", "s_branch")) else "s" if t.startswith("s_")
         else "ds" if t.startswith("ds_") else "vm")
    cnt[sec][k] += 1
for k in order:
    print("%-14s %s" % (k, dict(cnt[k])))
