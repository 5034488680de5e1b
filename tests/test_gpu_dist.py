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
