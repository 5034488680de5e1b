"""Command-line renderer: scene file in, PPM (and optionally the linear HDR buffer as PFM) out.

    python -m ptamd.cli scene.json [--spp N] [--out out.ppm] [--hdr-out out.pfm]
                                   [--chunk N] [--checkpoint state.npz] [--resume]
    python -m torch.distributed.run --nproc-per-node G -m ptamd.cli scene.json ...     (G GPUs of one node)

Replaces the reference's workflow of editing src/main.cu and recompiling.  Samples are added in
chunks, so a render can be checkpointed and resumed; with several ranks every rank renders its share
of the sample range and rank 0 combines the sums (one RCCL sum-reduce) and writes the image.
"""
import argparse
import os
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ptamd.cli")
    ap.add_argument("scene")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: the scene file's)")
    ap.add_argument("--out", default="output.ppm")
    ap.add_argument("--hdr-out", default="", help="also write the normalised linear radiance as PFM")
    ap.add_argument("--chunk", type=int, default=256, help="samples added per kernel launch")
    ap.add_argument("--checkpoint", default="", help="write the accumulation state here after every chunk (single GPU)")
    ap.add_argument("--resume", action="store_true", help="continue from --checkpoint if it exists")
    args = ap.parse_args(argv)
    if args.chunk < 1:
        ap.error("--chunk must be at least 1")
    if args.spp < 0:
        ap.error("--spp must not be negative")

    import ptamd
    from ptamd import dist as ptdist
    from ptamd import io as ptio
    from ptamd.progressive import Accumulator
    from ptamd.scenefile import description_hash, load_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tdist = None
    if world > 1:
        import torch
        import torch.distributed as tdist

        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    desc, settings = load_scene(args.scene)
    spp = ptdist.effective_spp(args.spp if args.spp else settings["spp"])
    scene, cam = ptamd.scene_from_description(desc, device=local_rank)
    begin, end = ptdist.shard_samples(spp, rank, world)
    # a checkpoint is only valid for the scene it was rendered from: the hash of the loaded description (the meshes, maps and
    # environment map the scene file names included, not just the file's own text) travels in the checkpoint
    tag = "scene-sha256:" + description_hash(desc)
    acc = Accumulator(scene, cam, seed=settings["seed"], device=local_rank, first_sample=begin, tag=tag)
    if args.resume and args.checkpoint and world == 1 and os.path.exists(args.checkpoint):
        acc.load(args.checkpoint, max_samples=end - begin)
        print(f"resumed at {acc.samples_done} samples per pixel", file=sys.stderr)
    t0 = time.time()
    while acc.samples_done < end - begin:
        acc.add(min(args.chunk, end - begin - acc.samples_done))
        if args.checkpoint and world == 1:
            acc.save(args.checkpoint)
    total = acc.radiance_sum()
    if world > 1:
        import torch

        t = torch.from_numpy(total).to(f"cuda:{local_rank}")
        ptdist.reduce_image(t, tdist, dst=0)
        total = t.cpu().numpy()
    if rank == 0:
        print(f"{spp} spp at {cam.res_x}x{cam.res_y} in {time.time() - t0:.2f} s on {world} GPU(s)", file=sys.stderr)
        ptio.save_ppm(args.out, ptamd.post_chain_host(total, spp, device=local_rank))
        if args.hdr_out:
            ptio.save_pfm(args.hdr_out, total / np.float32(spp))
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
