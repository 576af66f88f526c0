#!/usr/bin/env python3
"""Per-launch medians of every counter rocprofv3 collected for the step kernels under <dir>/<shape>/<pass>/ (the layout
tools/profile_large.sh and tools/profile_round.sh write) -> JSON on stdout: {shape: {counter: median, ..., "kernel_us": median}}."""
import collections, csv, glob, json, os, statistics, sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    vals, durs = collections.defaultdict(list), []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"] or "step_ex_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if vals:
        out[os.path.basename(d)] = dict({k: statistics.median(v) for k, v in sorted(vals.items())},
                                        launches=min(len(v) for v in vals.values()), kernel_us_under_counters=statistics.median(durs))
json.dump(out, sys.stdout, indent=1)
