#!/bin/bash
# Memory-side fetch bytes of the render kernel by array: the FETCH_SIZE pass of run_profile.sh on the library and on the
# probe builds that keep one array in cache (make variant NAME=texel0 DEFS=-DPT_PROBE_TEXEL0, NAME=hdri0 DEFS=-DPT_PROBE_HDRI0).
# usage: profiles/fetch_attr.sh <tag> [bench args]   -> gpurun_out/fetch_<tag>/<variant>/...
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/fetch_$tag
mkdir -p $out
cd $root
for v in base texel0 hdri0; do
  if [ $v = base ]; then lib=$root/cuda-path-tracer_amd/lib/libptamd.so; else lib=$root/cuda-path-tracer_amd/lib/variants/libptamd_$v.so; fi
  export PTAMD_LIB=$lib
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/$v -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 $* > $out/${v}_bench.json 2> $out/$v.err || { echo "pass $v failed"; tail -5 $out/$v.err; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for v in ("base", "texel0", "hdri0"):
    f = glob.glob(f"{out}/{v}/**/*counter_collection.csv", recursive=True)[0]
    best = {}
    for r in csv.DictReader(open(f)):
        if "render_kernel<false" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            best[r["Dispatch_Id"]] = best.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    res[v] = max(best.values()) * 1024 * 2          # KB -> bytes, gfx950: 128-byte requests tallied as 64
b = res["base"]
line = json.load(open(f"{out}/base_bench.json"))
d = {"workload": line["config"]["workload"], "source_hash": line["library_source_hash"],
     "source": "rocprofv3 --pmc FETCH_SIZE passes of profiles/fetch_attr.sh: the library and the two probe builds that keep one array "
               "in cache; bytes per launch of the timed kernel, FETCH_SIZE x2 (gfx950)",
     "fetch_bytes_per_launch": b, "textures": b - res["texel0"], "hdri": b - res["hdri0"],
     "geometry_and_scratch": res["texel0"] + res["hdri0"] - b}
json.dump(d, open(f"{out}/fetch_by_array.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
