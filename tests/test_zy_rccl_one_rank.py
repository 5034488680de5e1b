"""The RCCL branch of bench.py on a one-GPU box.  Collected late (only the allocation-free entry point's test comes
after it): it depends on the collective library starting up, which no other GPU test does."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_collectives_over_rccl_with_one_rank():
    """The collective calls bench.py makes for N > 1 -- process group on the RCCL backend, the sum-reduce of the
    image, barrier, the timer / counter all-reduces, teardown -- executed on RCCL with a process group of one rank
    (PTAMD_FORCE_DIST=1), which is all a one-GPU box can host; the counters equal the plain single-GPU run's."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", "teapot", "--res", "96x56", "--spp", "64", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=dict(env, PTAMD_FORCE_DIST="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["backend"] == "nccl" and d["n_gpus"] == 1 and d["counters"]["paths"] == 96 * 56 * 64
    plain = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    d1 = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["backend"] is None and d1["counters"] == d["counters"]
