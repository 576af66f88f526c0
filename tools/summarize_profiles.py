#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof_*) into the small tracked files under profiles/.

    python tools/summarize_profiles.py <round-tag>      e.g. r01

kernel trace  -> profiles/<tag>_kernel_stats.csv (uavx kernels only) 
PMC passes    -> profiles/<tag>_pmc_summary.json: per-launch medians for the step kernel, with the
                 gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM (x2 for wide coalesced
                 streaming reads; FETCH_SIZE/WRITE_SIZE are in KiB)."""
import collections, csv, glob, json, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)

stats = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_kt", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    rows = list(csv.reader(open(stats[-1])))
    keep = [rows[0]] + [r for r in rows[1:] if "uavx" in r[0]]
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(keep)
    print("kernel stats:", [(r[0][:60], r[1], r[3]) for r in keep[1:]])

summary = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        vals, dur = collections.defaultdict(list), []
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta = dict(kernel=r["Kernel_Name"].split("(")[0], grid=int(r["Grid_Size"]), vgpr=int(r["VGPR_Count"]),
                            sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]))
        for k, v in vals.items():
            summary[k] = dict(median=statistics.median(v), n=len(v), source=os.path.relpath(f, ROOT))
        if dur:
            summary.setdefault("_meta", meta)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fetch_kib, write_kib = summary["FETCH_SIZE"]["median"], summary["WRITE_SIZE"]["median"]
    summary["hbm_traffic_bytes_per_launch"] = dict(
        read=fetch_kib * 1024 * 2, write=write_kib * 1024, total=fetch_kib * 1024 * 2 + write_kib * 1024,
        note="FETCH_SIZE x2 (gfx950 counts 128-B requests of a wide coalesced read at 64 B), WRITE_SIZE exact; KiB units")
with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)
print(json.dumps(summary.get("hbm_traffic_bytes_per_launch"), indent=1))
