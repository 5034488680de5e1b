"""world_size-2 CPU test (gloo) of the multi-GPU driver logic: the sample range is sharded over
ranks, each rank accumulates its shard, one sum-reduce combines them.  The per-rank renderer here
is the CPU oracle (tests may use it; the product path on GPUs is pt_render_accumulate)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
    import torch
    import torch.distributed as dist

    import oracle_api as O
    from ptamd import dist as ptdist
    from ptamd import scenes as S

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = S.cornellbox_scene(res=(32, 18))
    sc, cam = O.scene_from_description(desc)

    def render_fn(b, e):
        img, _ = sc.render(cam, 1337, b, e)
        return torch.from_numpy(img)

    total, (b, e) = ptdist.render_sharded(render_fn, 8, rank, world, dist)
    assert (b, e) == ptdist.shard_samples(8, rank, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, total.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single(tmp_path, oracle, scenes):
    import torch.multiprocessing as mp

    port = _free_port()
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc, cam = oracle.scene_from_description(scenes.cornellbox_scene(res=(32, 18)))
    want, _ = sc.render(cam, 1337, 0, 8)
    assert got.shape == want.shape
    # identical sample set; only the association of the final sums differs
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    half, _ = sc.render(cam, 1337, 0, 4)
    assert not np.allclose(got, half, rtol=1e-3)
