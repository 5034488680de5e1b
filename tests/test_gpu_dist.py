"""Two ranks, one process each, rendering their shards of the sample range with the HIP path and
combining them with one sum-reduce: the multi-GPU driver logic end to end on the one GPU of the
test box (both ranks use cuda:0; the collective runs over gloo)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_two_rank_gpu_render_equals_single(pt, scenes, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "sum.npy")
    spp = 64
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_gpu_worker.py"), out, str(spp)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log, _ = p.communicate(timeout=240)
        assert p.returncode == 0, log[-2000:]
    got = np.load(out)
    g, cam = pt.scene_from_description(scenes.teapot_scene(res=(96, 56)))
    want = pt.render_image(g, cam, spp, seed=1337, n_split=4)
    half = pt.render_image(g, cam, spp // 2, seed=1337, n_split=4)
    g.close()
    assert got.shape == want.shape
    # the same sample set; only the association of the last sums differs
    assert np.allclose(got, want, rtol=2e-5, atol=1e-5)
    assert not np.allclose(got, half, rtol=1e-3)


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 path end to end -- rank start-up without a launcher, process group, sample sharding, the
    sum-reduce, the max-over-ranks timer and counter all-reduces, the rank-0 JSON line -- with two ranks
    sharing the one GPU of the test box over gloo (on a multi-GPU node the same code runs over RCCL)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "teapot", "--res", "96x56",
           "--spp", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["backend"] == "gloo"
    assert d["config"]["samples_per_rank"] == 32 and d["config"]["spp"] == 64
    assert d["counters"]["paths"] == 96 * 56 * 64          # both ranks' counters, summed
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["roofline"]["bound"] == "valu-issue" and d["roofline"]["kernel_ms"] > 0
    # the scene is built once per node: rank 0 builds the teapot's octree, rank 1 loads the built scene
    assert d["scene_by_rank"]["octree_builds"] == [1, 0] and len(d["scene_by_rank"]["build_upload_s"]) == 2
    assert d["config"]["parallelism"] == "sample-shard x2"
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "teapot", "--res", "96x56", "--spp", "64",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["counters"] == d["counters"]


def test_bench_shards_tiles_when_samples_are_few(tmp_path):
    """Five ranks on a 32-spp render (fewer than 8 samples per rank): bench.py shards the image's tiles instead of the
    samples; the ranks' counters add up to the single-process run's, one octree is built in all."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--workload", "teapot", "--res", "96x56", "--spp", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "5", "--backend", "gloo"] + common, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 5 and d["config"]["parallelism"] == "tile-shard x5" and d["config"]["samples_per_rank"] == 32
    assert d["scene_by_rank"]["octree_builds"] == [1, 0, 0, 0, 0]
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common, env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["counters"] == d["counters"] and d1["counters"]["paths"] == 96 * 56 * 32
