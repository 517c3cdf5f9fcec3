#!/usr/bin/env python3
"""Table of the rocprofv3 --pmc passes written by tools/tile_mem_counters.sh: one row per counter, one column per
configuration (averages per launch of the measured kernel: k_tile, or k_copy for the copy reference)."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
cfgs = sorted({re.sub(r"_g\d+$", "", os.path.basename(d)) for d in glob.glob(f"{root}/*_g*") if os.path.isdir(d)},
              key=lambda s: (s.split("_")[0], s))
table = collections.defaultdict(dict)
times = {}
for cfg in cfgs:
    try:
        times[cfg] = open(f"{root}/{cfg}.time").read().strip().splitlines()[-1].split()[-1]
    except Exception:
        times[cfg] = "?"
    want = "k_copy" if cfg.endswith("copy") else "k_tile"
    for f in glob.glob(f"{root}/{cfg}_g*/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for k, (c, v) in acc.items():
            table[k][cfg] = v / c
print("counter averages per launch; columns:")
for i, c in enumerate(cfgs):
    print(f"  [{i}] {c}  ms_per_launch={times[c]}")
print()
hdr = "%-46s" % "counter" + "".join("%14s" % f"[{i}]" for i in range(len(cfgs)))
print(hdr)
for k in sorted(table):
    print("%-46s" % k + "".join("%14s" % (("%.4g" % table[k][c]) if c in table[k] else "-") for c in cfgs))
