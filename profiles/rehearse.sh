#!/bin/bash
# Rehearsals of the N > 1 paths of bench.py on a one-GPU box (not scaling figures): two ranks on gloo at the nominal size, RCCL
# with a group of one rank, five ranks on gloo on the 32-spp cornell box (takes the tile axis): profiles/rehearse.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline > gpurun_out/bench_gloo2_rehearsal_$tag.json 2> gpurun_out/bench_gloo2.err || { echo "gloo2 failed"; tail -3 gpurun_out/bench_gloo2.err; exit 1; }
PTAMD_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/bench_rccl_one_rank_$tag.json 2> gpurun_out/bench_rccl1.err || { echo "rccl1 failed"; tail -3 gpurun_out/bench_rccl1.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 5 --backend gloo --workload cornellbox --no-cpu-baseline > gpurun_out/bench_gloo5_tiles_cornellbox_$tag.json 2> gpurun_out/bench_gloo5.err || { echo "gloo5 failed"; tail -3 gpurun_out/bench_gloo5.err; exit 1; }
for f in bench_gloo2_rehearsal_$tag bench_rccl_one_rank_$tag bench_gloo5_tiles_cornellbox_$tag; do
  python3 -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('parallelism'), d.get('scene_by_rank'), d.get('backend'))"
done
