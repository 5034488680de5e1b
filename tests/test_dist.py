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


def test_shard_work_picks_samples_or_tiles():
    sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
    from ptamd import dist as ptdist

    # enough samples per rank: contiguous sample ranges that tile [0, spp), every tile on every rank
    for spp, world in ((2048, 8), (64, 8), (32, 4), (33, 3)):
        got = [ptdist.shard_work(spp, r, world) for r in range(world)]
        assert all(t == (0, 1) for _, t in got)
        assert got[0][0][0] == 0 and got[-1][0][1] == spp
        assert all(got[r][0][1] == got[r + 1][0][0] for r in range(world - 1))
        assert all(e - b >= 8 for (b, e), _ in got)
    # fewer than 8 samples per rank: all samples, interleaved tile shares
    for spp, world in ((32, 8), (32, 5), (8, 2)):
        got = [ptdist.shard_work(spp, r, world, sample_begin=3) for r in range(world)]
        assert all(s == (3, 3 + spp) for s, _ in got)
        assert [t for _, t in got] == [(r, world) for r in range(world)]
    with pytest.raises(ValueError):
        ptdist.shard_work(32, 2, 2)
