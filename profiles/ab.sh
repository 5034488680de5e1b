#!/bin/bash
# A/B of library variants on one GPU box: profiles/ab.sh "<bench args>" name1 name2 ...  (name "base" = lib/libptamd.so)
args=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
for round in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then lib=$root/cuda-path-tracer_amd/lib/libptamd.so; else lib=$root/cuda-path-tracer_amd/lib/variants/libptamd_$n.so; fi
  PTAMD_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > $root/gpurun_out/ab/$n.$round.json 2> $root/gpurun_out/ab/$n.$round.err || { echo "$n failed"; tail -3 $root/gpurun_out/ab/$n.$round.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$root/gpurun_out/ab/$n.$round.json')); print('$n', $round, d['value'], 'Mpaths/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])"
done
done
