#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name.
usage: pmc_summary.py <dir> [substring filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if filt and filt not in name: continue
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ssw::", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}, "n=%d" % len(next(iter(acc[k].values()))))
