"""bench.py's argument handling on a machine without a GPU: a rank count that disagrees with the launcher is an
error, and --gpus N without a launcher starts N child ranks (which here stop at "needs a GPU")."""
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_must_match_the_launcher():
    r = _run(["--gpus", "2"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in r.stderr
    r = _run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


def test_gpus_without_launcher_starts_the_ranks():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("covered by the GPU test of the two-rank bench")
    r = _run(["--gpus", "2", "--backend", "gloo"])
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2        # both ranks ran and failed loudly
