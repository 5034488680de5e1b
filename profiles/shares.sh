#!/bin/bash
# wave-time shares of the render kernel's blocks for each workload: profiles/shares.sh [workloads...]
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/shares
for w in ${@:-hcandersen teapot cornellbox dragon synthetic}; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $w > $root/gpurun_out/shares/$w.json 2> $root/gpurun_out/shares/$w.err || { echo "$w failed"; tail -3 $root/gpurun_out/shares/$w.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$root/gpurun_out/shares/$w.json')); s=d['schedule']; t=s['wave_time_share']
print('$w', d['value'], 'Mpaths/s', ' '.join('%s %.3f' % (k, t[k]) for k in ('ent','node','leaf','scheduler','round')), 'trips(M) e %d n %d l %d f %d' % tuple(s[k]['trips']>>20 for k in ('ent','node','leaf','front')), 'occ e %.2f n %.2f l %.2f f %.2f' % tuple(s[k]['occupancy'] for k in ('ent','node','leaf','front')), 'lanes per node trip: ' + ' '.join('%s %.1f' % kv for kv in s['node_trip_lanes'].items()))"
done
