"""Host sanitizers over the multithreaded octree builder (the GPU code cannot run under sanitizers on
this pool; the host side can): AddressSanitizer + UBSan, then ThreadSanitizer, on the teapot at the
reference's depth, checked against the reference-derived structure counts."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, REPO

CSRC = os.path.join(REPO, "cuda-path-tracer_amd", "csrc")
SRC = os.path.join(REPO, "tests", "sanitize", "octree_sanitize.cpp")
TEAPOT = os.path.join(REPO, "examples", "teapot", "utah.obj")


@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined", "-fsanitize=thread"])
def test_octree_builder_under_sanitizers(tmp_path, flags):
    exe = tmp_path / "octree_sanitize"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-pthread", flags, "-fno-sanitize-recover=all", "-I", CSRC,
           SRC, os.path.join(CSRC, "octree_build.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), TEAPOT, "6"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr
    got = [int(x) for x in r.stdout.split()]
    known = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))["teapot"]
    assert got[0] == known["n_nodes"] and got[1] == known["n_leaf_refs"] and got[2] == known["max_leaf"]
    assert got[3:10] == known["nodes_per_depth"]
