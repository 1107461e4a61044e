"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace run (its *kernel_trace.csv): the --stats average
includes the first, cold launch (page mapping, instruction cache), so this prints calls, the first launch, and mean / median /
min / max of the rest.
    python3 tools/k1_trace_summary.py <rocprof output dir> <kernel name substring> [out.json]"""
import csv, glob, json, os, statistics, sys

d, sub = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = []
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        dur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
dur = [x[1] for x in sorted(dur)]
warm = dur[1:] if len(dur) > 1 else dur
out = {"kernel": sub, "calls": len(dur), "first_launch_us": dur[0] / 1e3, "warm_launches": len(warm),
       "warm_mean_us": statistics.mean(warm) / 1e3, "warm_median_us": statistics.median(warm) / 1e3,
       "warm_min_us": min(warm) / 1e3, "warm_max_us": max(warm) / 1e3,
       "warm_stdev_us": (statistics.pstdev(warm) / 1e3) if len(warm) > 1 else 0.0,
       "source": "rocprofv3 --kernel-trace, per-dispatch End - Start; the first launch reported apart"}
print(json.dumps(out))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
