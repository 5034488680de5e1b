#!/bin/bash
# sweep of the render kernel's launch choices: profiles/sweep_quota.sh "<bench args>" "<kernel builds>" "<quotas>"
args=$1; builds=${2:-heavy light}; quotas=${3:-12 24 40 56}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/sweep
for b in $builds; do for q in $quotas; do
  PTAMD_KERNEL=$b PTAMD_QUORUM=$q timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > $root/gpurun_out/sweep/$b.$q.json 2> $root/gpurun_out/sweep/$b.$q.err || { echo "$b $q failed"; tail -3 $root/gpurun_out/sweep/$b.$q.err; exit 1; }
  python3 -c "import json; d=json.load(open('$root/gpurun_out/sweep/$b.$q.json')); print('$b', $q, d['config']['workload'].split()[0], d['value'], 'Mpaths/s', 'kernel', d['roofline']['kernel_ms'])"
done; done
