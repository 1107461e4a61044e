"""Turns a rocprofv3 --pmc SQ_* pass over tools/solve_once.py into profiles/<name>.json.
usage: python tools/pmc_sq_summary.py <dir> <out.json> <n_systems> [command the pass ran, for the record]"""
import csv, glob, json, os, sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
waves = {}
for r in csv.DictReader(open(f)):
    if "fx::" not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    waves[r["Kernel_Name"]] = int(r["Grid_Size"]) // 64
out = {
    "source": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY "
              "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -- python3 " + (sys.argv[4] if len(sys.argv) > 4 else "tools/solve_once.py " + sys.argv[3] + " 2"),
    "note": "per launch; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles summed over wavefronts",
    "kernels": {},
}
for k, cs in sorted(acc.items()):
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        d["valu_active_fraction_of_wave_lifetime"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in d and waves.get(k):
        d["valu_instructions_per_wavefront"] = d["SQ_INSTS_VALU"] / waves[k]
        d["lds_instructions_per_wavefront"] = d.get("SQ_INSTS_LDS", 0.0) / waves[k]
    out["kernels"][k] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
