#!/usr/bin/env python3
"""Mean per launch of every counter per kernel from rocprofv3 --pmc csv output directories.
usage: pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items() if any(t in k for t in ("k_raw", "k_tx", "k_enc", "k_rx"))}
print(json.dumps(out, indent=1))
