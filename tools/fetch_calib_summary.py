"""rocprofv3 --pmc FETCH_SIZE pass over tools/probes/fetch_calib.bin -> profiles/<tag>_fetch_calibration.json:
bytes read (known) / (FETCH_SIZE KiB x 1024) per access width.
usage: python tools/fetch_calib_summary.py <pmc_dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
known = 1 << 30
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/probes/fetch_calib.bin (1 GiB per launch, 4x the Infinity Cache)",
       "bytes_read_per_launch": known, "kernels": {}}
for k, v in sorted(acc.items()):
    kib = sum(v) / len(v)
    out["kernels"][k] = {"FETCH_SIZE_KiB": kib, "launches": len(v), "bytes_per_FETCH_SIZE_byte": known / (kib * 1024.0)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
