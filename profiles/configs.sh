#!/bin/bash
# All five benchmark configurations at their nominal settings, one bench line each: profiles/configs.sh  -> gpurun_out/cfg_<workload>.json
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
for w in cornellbox teapot dragon hcandersen synthetic; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cfg_$w.json 2> gpurun_out/cfg_$w.err || { echo "$w failed"; tail -3 gpurun_out/cfg_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/cfg_$w.json')); print(d['config']['workload'], d['value'], 'Mpaths/s', d['roofline']['kernel_ms'], 'ms', d['launch'])"
done
