"""One rank of the two-process GPU test (tests/test_gpu_dist.py): renders its shard of the sample range
with the HIP path on cuda:0 and takes part in the sum-reduce.  Both ranks share the one GPU of the test
box, so the reduce runs over gloo on host tensors; on a multi-GPU node bench.py does the same over RCCL
with device tensors."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))


def main():
    out_path, spp = sys.argv[1], int(sys.argv[2])
    import torch
    import torch.distributed as dist

    import ptamd
    from ptamd import dist as ptdist
    from ptamd import scenes as S

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = S.teapot_scene(res=(96, 56))
    scene, cam = ptamd.scene_from_description(desc, device=0)

    def render_fn(b, e):
        img = ptamd.render_image(scene, cam, e - b, seed=1337, n_split=4, sample_begin=b)
        return torch.from_numpy(np.ascontiguousarray(img))

    total, (b, e) = ptdist.render_sharded(render_fn, spp, rank, world, dist)
    assert (b, e) == ptdist.shard_samples(spp, rank, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, total.numpy())
    scene.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
