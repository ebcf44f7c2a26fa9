#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_gpu.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_bench.json           the bench line printed under the profiler
  profiles/<tag>_pmc.json             FETCH_SIZE / WRITE_SIZE per kernel and launch
  profiles/traffic.json               HBM bytes per launch keyed the way bench.py looks them up

HBM bytes follow the MI355X guide: counters are in KiB; WRITE_SIZE is exact for 16-byte
streaming stores; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950, so
the read side is doubled (an upper bound for narrower reads).
usage: tools/summarize_profiles.py <tag> <packets>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
packets = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
b = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, tag + "_bench.json"))


def short(name):
    if "k_tx<" in name:     # k_tx<UNIFORM, REGIME, FROM_MSG, PNC>: the message-input instantiation is its own line
        args = [a.strip() for a in name.split("k_tx<", 1)[1].split(">", 1)[0].split(",")]
        return "k_tx_msg" if len(args) > 2 and args[2] == "true" else "k_tx"
    for k in ("k_encode_bits", "k_raw_modulate"):
        if k in name:
            return k
    return None


pmc = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, "pmc_" + ctr, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == ctr:
                agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[ctr + "_KiB_per_launch"] = sum(v) / len(v)
            pmc[k][ctr + "_launches"] = len(v)
for k, d in pmc.items():
    if "FETCH_SIZE_KiB_per_launch" in d and "WRITE_SIZE_KiB_per_launch" in d:
        d["hbm_bytes_per_launch"] = 1024.0 * (2.0 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"])
json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)

tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for k, d in pmc.items():
    if "hbm_bytes_per_launch" in d:
        traffic["%s_54M_1500B_%dpk" % (k, packets)] = d["hbm_bytes_per_launch"]
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(pmc, indent=1, sort_keys=True))
