"""Histogram of per-launch durations of the dominant uavx kernel from a rocprofv3 --kernel-trace CSV directory.
usage: python tools/launch_hist.py <dir>"""
import csv, glob, sys, collections
import numpy as np
d = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step" in r["Kernel_Name"] and "uavx" in r["Kernel_Name"]:
            d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
d.sort()
dur = np.array([x[1] for x in d], float) / 1e3
dur = dur[len(dur) // 5:]          # steady state: the last four fifths
print(f"{len(dur)} launches: mean {dur.mean():.2f} median {np.median(dur):.2f} p10 {np.percentile(dur, 10):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} max {dur.max():.2f} us")
edges = np.arange(np.floor(dur.min()), np.ceil(np.percentile(dur, 99.5)) + 1, 0.5)
h, e = np.histogram(dur, bins=edges)
for c, lo in zip(h, e[:-1]):
    if c: print(f"  {lo:6.1f}  {c:6d}  {'#' * int(60 * c / h.max())}")
print(f"  mean of launches below the median + 0.5 us: {dur[dur < np.median(dur) + 0.5].mean():.2f}; share above median + 2 us: {(dur > np.median(dur) + 2).mean():.3f}, they add {(dur[dur > np.median(dur) + 2] - np.median(dur)).sum() / len(dur):.2f} us to the mean")
