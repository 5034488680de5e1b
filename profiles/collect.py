#!/usr/bin/env python3
"""Copies the judged files of one profiling run from gpurun_out/ into profiles/<round>/.

usage: profiles/collect.py <tag> <round-dir>      e.g.  profiles/collect.py r1j profiles/r1
Expects gpurun_out/prof_<tag>/ (profiles/run_profile.sh), gpurun_out/bench_default_<tag>.json and,
optionally, gpurun_out/cfg_<workload>.json for the five configurations.  Writes kernel_stats_<tag>.csv, bench_line_<tag>.json,
bench_default_<tag>.json, summary_<tag>.json, configs_<tag>.json and refreshes traffic.json / pmc_ratios.json (the two files
bench.py reads), both stamped with the SHA-256 of the library sources the profiled run reported in its bench line
(`library_source_hash`).  bench.py drops the two files when the stamp no longer matches the sources it runs with."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(out, exist_ok=True)
prof = os.path.join(root, "gpurun_out", f"prof_{tag}")


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return files[-1]


shutil.copy(newest(f"{prof}/trace/*/*kernel_stats.csv"), f"{out}/kernel_stats_{tag}.csv")
shutil.copy(f"{prof}/trace_bench.json", f"{out}/bench_line_{tag}.json")
if os.path.exists(os.path.join(root, "gpurun_out", f"bench_default_{tag}.json")):
    shutil.copy(os.path.join(root, "gpurun_out", f"bench_default_{tag}.json"), f"{out}/bench_default_{tag}.json")
with open(f"{out}/summary_{tag}.json", "w") as f:
    subprocess.run([sys.executable, os.path.join(root, "profiles", "summarize.py"), prof], stdout=f, check=True)

bench = json.load(open(f"{out}/bench_line_{tag}.json"))
workload = bench["config"]["workload"]
source_hash = bench["library_source_hash"]
# the headline workload's counter files have the plain names; the other configurations' carry the workload's name
suffix = "" if workload.split()[0] == "hcandersen" else "_" + workload.split()[0]


def timed_kernel_counters(passes):
    """Sums of each counter over the dispatches of the timed (non-counting) render kernel."""
    acc, name = collections.defaultdict(float), None
    for p in passes:
        for r in csv.DictReader(open(newest(f"{prof}/pmc_{p}/*/*counter_collection.csv"))):
            if "render_kernel<false" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                name = r["Kernel_Name"].split("(")[0].replace("void ptamd::", "")
    return acc, name


def dispatches(pass_name):
    """number of dispatches of the timed kernel in one pass (the counter sums are divided by it)"""
    ids = set()
    for r in csv.DictReader(open(newest(f"{prof}/pmc_{pass_name}/*/*counter_collection.csv"))):
        if "render_kernel<false" in r["Kernel_Name"]:
            ids.add(r.get("Dispatch_Id"))
    return max(len(ids), 1)


full, kname = timed_kernel_counters(["tcc1", "tcc2"])
fetch_kb, write_kb = full["FETCH_SIZE"] / dispatches("tcc1"), full["WRITE_SIZE"] / dispatches("tcc2")
traffic = int(fetch_kb * 1024 * 2 + write_kb * 1024)
json.dump({
    "workload": workload, "kernel": kname, "source_hash": source_hash,
    "source": f"rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum / --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum (separate passes) over `python3 bench.py "
              f"--no-cpu-baseline --workload {workload.split()[0]} --steps 1 --warmup 0`, profiles/run_profile.sh {tag}; per dispatch of the timed kernel",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "FETCH_SIZE x2 on gfx950 (128-byte requests tallied as 64 bytes); calibrated for streaming reads, this kernel "
                  "gathers: treat as +-2x",
    "traffic_bytes_per_launch": traffic}, open(f"{out}/traffic{suffix}.json", "w"), indent=1)

def timed_kernel_ms(pass_name):
    """mean duration of the timed kernel's dispatches in one counter pass (the CSV carries start / end per counter row)"""
    spans = {}
    for r in csv.DictReader(open(newest(f"{prof}/pmc_{pass_name}/*/*counter_collection.csv"))):
        if "render_kernel<false" in r["Kernel_Name"]:
            spans[r.get("Dispatch_Id")] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    return sum(spans.values()) / max(len(spans), 1)


acc, kname = timed_kernel_counters(["sq1", "sq2", "tcc1", "tcc2", "tcp"])
g = acc["GRBM_GUI_ACTIVE"] / 8
n_disp = dispatches("sq1")
lanes = acc["SQ_THREAD_CYCLES_VALU"] / (64 * acc["SQ_ACTIVE_INST_VALU"])
json.dump({
    "workload": workload, "kernel": kname, "source_hash": source_hash,
    "source": f"rocprofv3 --pmc passes of profiles/run_profile.sh {tag} (the benchmark's own launch: same scene, full sample count), "
              "counters of the timed kernel only",
    "valu_instructions_per_cycle_per_simd": round(acc["SQ_INSTS_VALU"] / (1024 * g), 4),
    "dispatches_per_pass": dispatches("sq1"),
    "valu_busy": round(acc["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g), 3),
    "active_lanes_per_valu_instruction": round(acc["SQ_THREAD_CYCLES_VALU"] / (64 * acc["SQ_ACTIVE_INST_VALU"]), 3),
    "wait_for_issue_over_wave_cycles": round(acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES"], 3),
    "waitcnt_over_wave_cycles": round(acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES"], 3),
    "l2_hit_rate": round(acc["TCC_HIT_sum"] / acc["TCC_REQ_sum"], 3),
    "l1_miss_rate": round(acc["TCP_TCC_READ_REQ_sum"] / acc["TCP_TOTAL_CACHE_ACCESSES_sum"], 3),
    "waves_per_simd": int(kname.split(",")[1].strip(" >")),
    # what the roofline line is computed from: vector instructions of one launch, the lanes active in them, and the pass's own clock
    "valu_instructions_per_launch": acc["SQ_INSTS_VALU"] / n_disp,
    "salu_instructions_per_launch": acc["SQ_INSTS_SALU"] / n_disp,
    "valu_lane_ops_per_launch": acc["SQ_INSTS_VALU"] / n_disp * 64 * lanes,
    "kernel_ms_in_pass": {p: round(timed_kernel_ms(p), 3) for p in ("sq1", "tcp")},
    "cycle_base": "GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)",
    "cycles_per_launch": g / dispatches("tcp"),
    "effective_clock_GHz": round(g / dispatches("tcp") / (timed_kernel_ms("tcp") * 1e6), 3),
    "raw": {k: acc[k] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_SALU",
                                "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum",
                                "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum") if k in acc}},
    open(f"{out}/pmc_ratios{suffix}.json", "w"), indent=1)

cfgs = []
for w in () if suffix else ("cornellbox", "teapot", "dragon", "hcandersen", "synthetic"):
    p = os.path.join(root, "gpurun_out", f"cfg_{w}.json")
    if os.path.exists(p):
        cfgs.append(json.load(open(p)))
if cfgs:
    json.dump(cfgs, open(f"{out}/configs_{tag}.json", "w"), indent=1)

print("traffic TB", traffic / 1e12, json.load(open(f"{out}/pmc_ratios{suffix}.json")))
for line in open(f"{out}/kernel_stats_{tag}.csv"):
    if "ptamd" in line:
        print(line.strip()[:140])
for d in cfgs:
    print(d["config"]["workload"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["post_chain_ms"], d["scene_build_upload_s"], d["launch"])
print("traced bench", bench["value"], bench["roofline"]["kernel_ms"])

# memory-side fetch bytes by array (profiles/fetch_attr.sh <tag>), if that run exists
fa = os.path.join(root, "gpurun_out", f"fetch_{tag}", "fetch_by_array.json")
if os.path.exists(fa) and not suffix:
    shutil.copy(fa, f"{out}/fetch_by_array.json")
