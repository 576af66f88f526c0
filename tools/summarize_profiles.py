#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof/<shape>/<pass>/, written by tools/profile_round.sh) into the small
tracked files under profiles/.

    python tools/summarize_profiles.py <round-tag>      e.g. r02

kernel trace  -> profiles/<tag>_kernel_stats_<shape>.csv (uavx kernels only)
PMC passes    -> profiles/<tag>_pmc_summary.json: {"kernels": [one entry per profiled shape], "calibration": {...}}.
                 Every entry holds the per-launch MEDIAN of each raw counter for the dominant step kernel, and
                 hbm_traffic_bytes_per_launch = {read, write, total, read_raw, factor}: FETCH_SIZE / WRITE_SIZE are KiB;
                 the read side is multiplied by the factor the calibration run (tools/micro/fetch_calib.hip: known byte
                 counts in this kernel's own access widths) measured for it; raw values stay next to the corrected ones.
"""
import collections, csv, glob, json, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
PROF = os.path.join(ROOT, "gpurun_out", "prof")
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
meta_run = json.load(open(os.path.join(PROF, "meta.json"))) if os.path.exists(os.path.join(PROF, "meta.json")) else {}


def counter_rows(path):
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))


def calibration():
    """bytes moved per launch / bytes the counter reports, per access shape (kernel name)."""
    d = os.path.join(PROF, "calib")
    if not os.path.isdir(d):
        return None
    known = 256 * 2 ** 20
    res = collections.defaultdict(dict)
    for p in ("fetch", "write", "tcc_ea", "tcc_hit"):
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in counter_rows(os.path.join(d, p)):
            k = r["Kernel_Name"]
            if "read_kernel" in k or "write_kernel" in k:
                vals[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in vals.items():
            for c, v in cs.items():
                res[k][c] = statistics.median(v)
    for k, cs in res.items():
        if "FETCH_SIZE" in cs and "read_kernel" in k:
            cs["true_bytes_over_FETCH_SIZE"] = known / (cs["FETCH_SIZE"] * 1024) if cs["FETCH_SIZE"] else None
        if "WRITE_SIZE" in cs and "write_kernel" in k:
            cs["true_bytes_over_WRITE_SIZE"] = known / (cs["WRITE_SIZE"] * 1024) if cs["WRITE_SIZE"] else None
    return dict(known_bytes_per_launch=known, by_kernel=res)


calib = calibration()


def read_factor():
    """FETCH_SIZE correction for this kernel's reads (8- and 16-byte per lane streams); 2.0 (the guide's figure for
    wide coalesced reads) when no calibration run is present."""
    if not calib:
        return 2.0, "MI355X_MICROARCH.md §HBM (x2 for wide coalesced reads); no calibration run found"
    f = [v["true_bytes_over_FETCH_SIZE"] for k, v in calib["by_kernel"].items()
         if "read_kernel" in k and v.get("true_bytes_over_FETCH_SIZE") and ("float2" in k or "float4" in k or "double2" in k or "HIP_vector_type" in k)]
    if not f:
        return 2.0, "calibration run had no usable read kernels"
    return statistics.median(f), "tools/micro/fetch_calib.hip: median of known bytes / FETCH_SIZE over the 8- and 16-byte-per-lane read kernels"


entries = []
for d in sorted(glob.glob(os.path.join(PROF, "*"))):
    shape = os.path.basename(d)
    if not os.path.isdir(d) or shape == "calib":
        continue
    stats = sorted(glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.reader(open(stats[-1])))
        keep = [rows[0]] + [r for r in rows[1:] if "uavx" in r[0]]
        with open(os.path.join(out_dir, f"{tag}_kernel_stats_{shape}.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
        print(shape, "kernel stats:", [(r[0][:50], r[1], r[3]) for r in keep[1:3]])
    summary, meta = {}, None
    for p in ("fetch", "write", "tcc_hit", "tcc_ea", "sq", "sq2"):
        vals, durs = collections.defaultdict(list), []
        for r in counter_rows(os.path.join(d, p)):
            if "step_kernel" in r["Kernel_Name"] or "step_ex_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta = dict(kernel=r["Kernel_Name"].split("(")[0], grid=int(r["Grid_Size"]), vgpr=int(r["VGPR_Count"]),
                            sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), shape=shape, **meta_run)
        for k, v in vals.items():
            summary[k] = dict(median=statistics.median(v), n=len(v))
    if not summary:
        continue
    summary["_meta"] = meta
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        fk, wk = summary["FETCH_SIZE"]["median"], summary["WRITE_SIZE"]["median"]
        fac, why = read_factor()
        summary["hbm_traffic_bytes_per_launch"] = dict(
            read_raw=fk * 1024, factor=fac, read=fk * 1024 * fac, write=wk * 1024, total=fk * 1024 * fac + wk * 1024,
            total_uncorrected=fk * 1024 + wk * 1024, note=f"FETCH_SIZE x {fac:.3f} ({why}); WRITE_SIZE as reported; counters are KiB")
    if "TCC_HIT_sum" in summary and "TCC_MISS_sum" in summary:
        h, m = summary["TCC_HIT_sum"]["median"], summary["TCC_MISS_sum"]["median"]
        summary["l2_hit_rate"] = h / (h + m) if h + m else None
    entries.append(summary)
    print(shape, json.dumps(summary.get("hbm_traffic_bytes_per_launch")), "L2 hit", summary.get("l2_hit_rate"))

with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(dict(kernels=entries, calibration=calib, **meta_run), f, indent=1, sort_keys=True)
if calib:
    for k, v in sorted(calib["by_kernel"].items()):
        print(k[:70], {c: round(x, 3) if isinstance(x, float) else x for c, x in v.items()})
