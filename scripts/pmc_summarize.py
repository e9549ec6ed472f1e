#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel name:  scripts_pmc_summarize.py gpurun_out/pmc_<tag>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    if not k.startswith("rd::"): continue
    print(k)
    for c in sorted(agg[k]):
        n = calls[k][c]
        print(f"   {c:32s} total {agg[k][c]:.4g}  per-dispatch {agg[k][c]/n:.4g}  (n={n})")
