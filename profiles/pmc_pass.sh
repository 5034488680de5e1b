#!/bin/bash
# one PMC pass: profiles/pmc_pass.sh <tag> "<counters>" [bench args...]; prints per-kernel sums of each counter
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd $root
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.txt || { echo "pass failed"; tail -5 $out/err.txt; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ptamd" not in k: continue
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:40s} sum={v:.4g} dispatches={n}")
PY
