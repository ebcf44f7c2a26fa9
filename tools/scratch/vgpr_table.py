#!/usr/bin/env python3
"""VGPRs / occupancy of every instantiation of one kernel, from hipcc's resource-usage remarks.
usage: vgpr_table.py <file.hip> <kernel-name-substring> [extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, name = sys.argv[1], sys.argv[2]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
       "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "ofdm_amd", "csrc"), "-c", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", src] + sys.argv[3:]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
for line in err.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    m = re.search(r"remark:\s+VGPRs: (\d+)", line)
    if m and cur and name in cur:
        args = re.findall(r"L[ib](\d+)E", cur.split(name)[1].split("EvNS")[0])
        print("%s<%s> %s" % (name, ",".join(args), m.group(1)))
