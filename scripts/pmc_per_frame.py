#!/usr/bin/env python3
"""Per-FRAME sums of rocprofv3 --pmc passes made with scripts/pmc_frames.py: every counter summed over all rd:: dispatches of the
run, divided by the number of frames; per kernel as well.  usage: pmc_per_frame.py <dir with one sub-directory per pass> <frames>"""
import csv, glob, json, os, sys, collections
root, frames = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(float)
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("rd::"):
            continue
        v = float(row["Counter_Value"])
        tot[row["Counter_Name"]] += v
        per[k][row["Counter_Name"]] += v
        cnt[k][row["Counter_Name"]] += 1
out = {"frames": frames, "per_frame": {c: tot[c] / frames for c in sorted(tot)}}
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 128-B requests at 64 B for 16-B-per-lane reads -> x2; KiB units; includes Infinity-Cache hits
    out["hbm_side_bytes_per_frame"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / frames
print(json.dumps(out, indent=1))
for k in sorted(per):
    print(k)
    for c in sorted(per[k]):
        print(f"   {c:34s} per frame {per[k][c] / frames:.5g}   per dispatch {per[k][c] / cnt[k][c]:.5g}  (dispatches per frame {cnt[k][c] / frames:.1f})")
