#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(stderr of the compile, e.g. `make -C ofdm_amd/csrc asm 2> usage.txt`); with two files: the kernels
whose figures differ."""
import re
import sys


def table(path):
    t = open(path).read()
    out = {}
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split("\n")[0].split(" ")[0]
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        out[name] = (g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
    return out


if __name__ == "__main__":
    a = table(sys.argv[1])
    if len(sys.argv) > 2:
        b = table(sys.argv[2])
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print(k[:90], a.get(k), "->", b.get(k))
    else:
        print("kernel  (VGPRs, SGPRs, scratch B/lane, waves/SIMD, static LDS B)")
        for k, v in a.items():
            print(k[:90], v)
