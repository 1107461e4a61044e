"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<name>.json.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> <n_systems>
Unit handling per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KiB; on gfx950 FETCH_SIZE tallies 128-B read requests as 64 B, so read bytes = 2 x FETCH_SIZE x 1024;
WRITE_SIZE x 1024 is exact."""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 tools/solve_once.py "
              + sys.argv[4] + " 2",
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM) gfx950 FETCH_SIZE tallies 128-B read requests as "
            "64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact",
    "n_systems": int(sys.argv[4]),
    "kernels": {},
}
for k in sorted(fetch):
    if "fx::" not in k:
        continue
    out["kernels"][k] = {
        "launches_averaged": nf[k],
        "FETCH_SIZE_KiB_per_launch": fetch[k],
        "WRITE_SIZE_KiB_per_launch": write.get(k, 0.0),
        "hbm_bytes_per_launch": int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024),
    }
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
