#!/usr/bin/env python3
"""Summarises a profiles/run_profile.sh output directory: per-kernel durations from the kernel
trace and the PMC counters of the render kernel, as one JSON document."""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r


def main(d):
    out = {"kernels": {}, "pmc": {}}
    for r in rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        k = r["Kernel_Name"].split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        e = out["kernels"].setdefault(k, {"calls": 0, "total_ms": 0.0, "vgpr": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"),
                                          "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size")})
        e["calls"] += 1
        e["total_ms"] += dur
    for e in out["kernels"].values():
        e["avg_ms"] = e["total_ms"] / e["calls"]
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
            k = r["Kernel_Name"].split("(")[0]
            if "render_kernel" not in k:
                continue
            c = out["pmc"].setdefault(r["Counter_Name"], {"sum": 0.0, "dispatches": 0})
            c["sum"] += float(r["Counter_Value"])
            c["dispatches"] += 1
    for c in out["pmc"].values():
        c["per_dispatch"] = c["sum"] / max(c["dispatches"], 1)
    for name in ("trace_bench.json",):
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                out["bench"] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception:
                pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
