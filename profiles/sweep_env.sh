#!/bin/bash
# sweep of scheduling knobs read from the environment at pt_scene_commit:
#   profiles/sweep_env.sh "<bench args>" "<kernel builds>" "<quorums>" "<leaf quotas>"
args=$1; builds=${2:-heavy}; quotas=${3:-40}; leafs=${4:-12}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/sweep
for b in $builds; do for q in $quotas; do for l in $leafs; do
  tag=$b.q$q.l$l
  PTAMD_KERNEL=$b PTAMD_QUORUM=$q PTAMD_LEAF_QUOTA=$l timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > $root/gpurun_out/sweep/$tag.json 2> $root/gpurun_out/sweep/$tag.err || { echo "$tag failed"; tail -3 $root/gpurun_out/sweep/$tag.err; exit 1; }
  python3 -c "import json; d=json.load(open('$root/gpurun_out/sweep/$tag.json')); s=d['schedule']; print('$tag', d['config']['workload'].split()[0], d['value'], 'Mpaths/s', 'kernel', d['roofline']['kernel_ms'], 'occ node %.3f leaf %.3f front %.3f ent %.3f' % (s['node']['occupancy'], s['leaf']['occupancy'], s['front']['occupancy'], s['ent']['occupancy']), 'Mtrips n %d l %d f %d e %d' % (s['node']['trips']>>20, s['leaf']['trips']>>20, s['front']['trips']>>20, s['ent']['trips']>>20))"
done; done; done
