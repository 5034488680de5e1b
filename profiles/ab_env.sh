#!/bin/bash
# A/B of environment settings on one library: profiles/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...  (each entry may hold several assignments)
args=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > $root/gpurun_out/ab/env$i.json 2> $root/gpurun_out/ab/env$i.err || { echo "$setting failed"; tail -3 $root/gpurun_out/ab/env$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('$root/gpurun_out/ab/env$i.json')); print('$setting', d['value'], 'Mpaths/s', 'kernel', d['roofline']['kernel_ms'], 'node occ %.3f leaf %.3f front %.3f rounds %d' % (d['schedule']['node']['occupancy'], d['schedule']['leaf']['occupancy'], d['schedule']['front']['occupancy'], d['schedule']['front']['trips']))"
done
