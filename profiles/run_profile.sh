#!/bin/bash
# Profiles the benchmark's render kernel on the GPU box: one kernel-trace pass (durations) and
# separate PMC passes (counters never share a run with trace options other than --kernel-trace).
# usage: profiles/run_profile.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd $root
# the kernel-trace pass runs the benchmark's default command line, so its per-kernel average is
# directly comparable with the HIP-event time in the bench line; the counter passes run the same
# workload (the same bench arguments) at its full sample count (the launch the bench times), one timed step (PMC_ARGS overrides)
args="--no-cpu-baseline $* ${PMC_ARGS:---steps 1 --warmup 0}"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $* > $out/trace_bench.json 2> $out/trace.err || { echo "trace pass failed"; tail -5 $out/trace.err; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py $args > $out/pmc_${name}_bench.json 2> $out/pmc_$name.err || { echo "pmc pass $name failed"; tail -5 $out/pmc_$name.err; exit 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
pass tcc1 FETCH_SIZE TCC_HIT_sum
# (memory-side traffic for roofline.traffic: FETCH_SIZE of pass tcc1 and WRITE_SIZE of pass tcc2 -- they cannot share a pass)
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
echo "profile passes done: $out"
