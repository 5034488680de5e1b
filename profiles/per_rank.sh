#!/bin/bash
# What one rank of an N-GPU run of the benchmark does (2048 / N samples of every pixel), measured on one GPU: profiles/per_rank.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/per_rank
for n in 1 2 4 8 16; do
  spp=$((2048 / n))
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --spp $spp --steps 3 > gpurun_out/per_rank/n$n.json 2> gpurun_out/per_rank/n$n.err || { echo "n=$n failed"; tail -3 gpurun_out/per_rank/n$n.err; exit 1; }
done
python3 - <<'PY' > gpurun_out/per_rank_workloads_$tag.json
import json
rows = []
for n in (1, 2, 4, 8, 16):
    d = json.load(open(f"gpurun_out/per_rank/n{n}.json"))
    rows.append(dict(ranks=n, samples_per_rank=2048 // n, ms_per_step=d["ms_per_step"], kernel_ms=d["roofline"]["kernel_ms"], mpaths_per_s=d["value"]))
base = rows[0]["mpaths_per_s"]
for r in rows:
    r["rate_vs_one_gpu"] = round(r["mpaths_per_s"] / base, 4)
print(json.dumps(dict(source="profiles/per_rank.sh: bench.py --spp 2048/N on ONE GPU -- the render of one rank of an N-GPU run, without the final reduce; not a scaling measurement", rows=rows), indent=1))
PY
cat gpurun_out/per_rank_workloads_$tag.json
