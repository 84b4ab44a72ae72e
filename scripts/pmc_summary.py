"""Aggregate rocprofv3 --pmc counter_collection.csv files: per ppm:: kernel, per counter: sum over dispatches,
dispatch count.  Usage: pmc_summary.py <dir> [--meta key=value ...] > summary.json   (raw CSVs can then be deleted)"""
import csv, glob, json, os, sys
from collections import defaultdict

def main(root, meta=None):
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "ppm::" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                c = row["Counter_Name"]; v = float(row["Counter_Value"])
                out[k][c][0] += v; out[k][c][1] += 1
    res = {k: {c: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for c, v in d.items()} for k, d in out.items()}
    # kernel durations from the traces
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "ppm::" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                dur[k][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6; dur[k][1] += 1
    for k, v in dur.items():
        res.setdefault(k, {})["_duration_ms_all_passes"] = {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)}
    if meta:
        res["_meta"] = meta
    json.dump(res, sys.stdout, indent=1, sort_keys=True)

if __name__ == "__main__":
    meta = {}
    if "--meta" in sys.argv:
        for kv in sys.argv[sys.argv.index("--meta") + 1:]:
            k, _, v = kv.partition("=")
            meta[k] = int(v) if v.isdigit() else v
    main(sys.argv[1], meta)
