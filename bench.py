#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s and time-to-2048-spp on the hcandersen-class scene at 1920x1080.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full render of the workload: every pixel's sample range [0, spp) traced by the
persistent HIP kernel (pt_render_accumulate through the C ABI), split buffers compounded on the
device and -- for N > 1 -- the per-rank radiance sums combined on rank 0 with one RCCL sum-reduce.
With N ranks the sample range is sharded (rank r renders spp/N samples of every pixel), so total
work is fixed: strong scaling.  The scene (procedural stand-ins for the reference's unshipped
80k.obj, textures and HDRI) is resident in HBM before the timed region starts; scene generation,
octree build, upload and post-processing are reported separately and are not part of `value`.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself as child
processes.  `--backend gloo` (or PTAMD_DIST_BACKEND=gloo) runs the collective over gloo on host
copies and lets several ranks share one GPU: a rehearsal of the N > 1 path on a one-GPU box.
PTAMD_FORCE_DIST=1 makes a single rank set up its process group and run the same collective calls
(over RCCL with the default backend), which is how a one-GPU box exercises that branch.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak
# algorithmic bytes per unit of traversal / shading work (what one unit must read from the scene arrays)
B_NODE = 8                   # one flattened octree node record
B_TRI = 40                   # one pre-gathered triangle record (v0, e1, e2, id), stored two to an 80-byte pair
B_SHADE = 64 + 24            # per hit: one shading record (3 normals + 3 uvs) + re-read of the accepted triangle's e1, e2
B_TEXEL = 3                  # RGB8 texel (gloss: 1, counted as 3 for simplicity -> upper bound below 0.1 % of the total)
B_ENV = 48                   # bilinear HDRI lookup: 4 texels x 12 B
B_ENTITY = 128               # entity record (112 B, counted as one 128-byte line) per (segment, entity)


N_SIMD = 1024                  # 256 CUs x 4 SIMDs
NOMINAL_GHZ = 2.4
# ONE vector peak: a SIMD retires 32 lanes of one vector instruction per cycle (a wave64 instruction in two cycles), so the
# chip offers 1024 x 32 x 2.4e9 lane-operations per second; counted as fused multiply-adds that is the 157.3 TFLOP/s
# fp32 vector peak.  Both fractions of the roofline line are taken against it.
LANE_OP_PEAK_T = N_SIMD * 32 * NOMINAL_GHZ / 1e3       # = 78.6 T lane-op/s
VALU_PEAK_TFLOPS = 2 * LANE_OP_PEAK_T                  # = 157.3 TFLOP/s
PROFILE_DIR = "r3"             # profiles/<dir>/ holds the counter passes this line may quote


def launch_ranks(n):
    """Starts ranks 0..n-1 of this script as child processes (before this process touches a GPU) and
    returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def stamped_profile(name, workload_name, world):
    """Counter results of separate rocprofv3 --pmc passes over this benchmark (profiles/run_profile.sh +
    profiles/collect.py write them).  They are only reported when they were measured on this very workload
    with the library sources as they are now (SHA-256 of csrc/ + the header): otherwise None."""
    import ptamd

    if world != 1:
        return None
    wl = workload_name.split()[0]
    if wl != "hcandersen":          # the headline workload's files have the plain names
        name = name.replace(".json", f"_{wl}.json")
    try:
        with open(os.path.join(REPO, "profiles", PROFILE_DIR, name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("workload") != workload_name or not d.get("source_hash") or d.get("source_hash") != ptamd.source_hash():
        return None
    d["_file"] = f"profiles/{PROFILE_DIR}/" + name
    return d


def algorithmic_flops(counters, n_mesh_entities):
    """Useful fp32 operations of one launch: 50 per triangle test, 25 per node visit, per segment a 24-flop slab test
    for every mesh entity and about 150 of shading."""
    return (counters["tri_tests"] * 50 + counters["node_visits"] * 25
            + counters["segments"] * (24 * n_mesh_entities + 150))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hcandersen", choices=["hcandersen", "teapot", "cornellbox", "dragon", "synthetic"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (default: the config's)")
    ap.add_argument("--res", type=str, default="", help="override resolution WxH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save-ppm", type=str, default="", help="write the post-processed image of the last step")
    ap.add_argument("--backend", default=os.environ.get("PTAMD_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo: host copies, ranks may share a GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: run the ranks as children of this process (which has not initialised any GPU)
        sys.exit(launch_ranks(args.gpus))

    # The one JSON line goes to the real stdout; everything else that writes to file descriptor 1 from here on (RCCL's and
    # gloo's start-up banners, which come from native code) goes to stderr.
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import ptamd
    from ptamd import dist as ptdist
    from ptamd import scenes as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., or without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} GPU(s): RCCL needs one GPU per rank (use --backend gloo to rehearse on fewer)")
    dev = local_rank % n_dev
    torch.cuda.set_device(dev)
    # PTAMD_FORCE_DIST=1: a process group even for one rank, so that the collective calls of the N > 1 path (RCCL
    # reduce, barrier, all_reduce) run on a box with a single GPU
    use_dist = world > 1 or os.environ.get("PTAMD_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def progress(msg):
        # progress lines on stderr (never inside the timed region): a run of the large configurations is silent for minutes otherwise
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    cfg = S.CONFIGS[args.workload]
    res = tuple(int(x) for x in args.res.split("x")) if args.res else cfg["res"]
    spp = ptdist.effective_spp(args.spp if args.spp else cfg["spp"])
    # The scene is generated, and its octrees are built, ONCE per node: by rank 0, which hands the built scene to the other
    # ranks through a file on the RAM-backed /dev/shm (pt_scene_save / pt_scene_load); they only upload it.
    desc, t_gen, t0 = None, 0.0, time.time()
    share = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"ptamd_bench_{os.environ.get('MASTER_PORT', '0')}_{args.workload}.scene")
    if rank == 0:
        desc = cfg["builder"](res=res)
        t_gen = time.time() - t0
        progress(f"scene description made in {t_gen:.1f} s")
        t0 = time.time()
        scene, cam = ptamd.scene_from_description(desc, device=dev)
        meshes = [e for e in desc["entities"] if e["kind"] == "mesh"]
        meta = {"camera": desc["camera"], "n_entities": len(desc["entities"]), "n_mesh": len(meshes),
                "depth": max([e.get("octree_depth", 0) for e in meshes] or [0]), "triangles": int(sum(len(e["tri"]) for e in meshes))}
        if world > 1:
            scene.save(share)
    if world > 1:
        box = [meta if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)          # (also the signal that the file is complete)
        meta = box[0]
        if rank != 0:
            t0 = time.time()
            scene = ptamd.load_scene(share, device=dev)
            c = meta["camera"]
            cam = ptamd.make_camera(c["position"], c["direction"], c["res"], c["focal_length"], c["aperture"], c["focus_distance"])
        dist.barrier()
        if rank == 0 and os.path.exists(share):
            os.remove(share)
    t_upload = time.time() - t0
    progress(f"scene built and uploaded in {t_upload:.1f} s")
    n_pixels = res[0] * res[1]
    n_split = 4
    n_entities = meta["n_entities"]
    (s_begin, s_end), tiles = ptdist.shard_work(spp, rank, world)

    split = torch.zeros((n_split, res[1], res[0], 3), dtype=torch.float32, device=f"cuda:{dev}")
    local_sum = torch.zeros((res[1], res[0], 3), dtype=torch.float32, device=f"cuda:{dev}")
    stream = torch.cuda.current_stream().cuda_stream
    L = ptamd.lib()

    def step(want_counters=False):
        split.zero_()
        cnt = scene.render_accumulate(cam, 1337, s_begin, s_end, n_split, split.data_ptr(), want_counters, stream, tiles=tiles)
        ptamd.check(L.pt_compound_buffers(dev, split.data_ptr(), local_sum.data_ptr(), n_pixels, n_split, stream))
        if use_dist:
            if args.backend == "nccl":
                dist.reduce(local_sum, dst=0, op=dist.ReduceOp.SUM)
            else:
                host = local_sum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    local_sum.copy_(host)
        return cnt

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    launch = scene.launch()      # the render kernel's launch choices (kernel build, quorum) picked at commit
    for i in range(args.warmup):
        step()
        progress(f"warm-up step {i + 1} of {args.warmup} done")
    # one untimed pass with the in-kernel work counters (identical work: same seed, same samples)
    counters = step(want_counters=True)
    torch.cuda.synchronize()
    progress("counting pass done")
    schedule = scene.schedule_counters()
    schedule["wave_time_share"] = scene.block_cycles()
    nt = max(schedule["node"]["trips"], 1)
    schedule["node_trip_lanes"] = dict(walking=schedule["node"]["lanes"] / nt, **{k: v / nt for k, v in scene.node_trip_idle_lanes().items()})
    scene.set_profiling(True)
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # the event pair sits on the launch stream; reading it waits only for that step's kernel
        kernel_ms.append(scene.last_render_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    scene.set_profiling(False)
    progress(f"{args.steps} timed steps done in {elapsed:.2f} s")

    cdev = f"cuda:{dev}" if args.backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    cvec = torch.tensor([counters[k] for k in ptamd.COUNTER_NAMES], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cvec, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    # per rank: seconds from scene description (rank 0) or scene file (others) to the committed scene, octrees built
    build_info = [(round(t_upload, 3), ptamd.octree_builds())]
    if use_dist and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, build_info[0])
        build_info = gathered
    total_counters = {k: int(v) for k, v in zip(ptamd.COUNTER_NAMES, cvec.tolist())}

    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
    paths_per_step = n_pixels * spp
    value = paths_per_step / (ms_per_step / 1000.0) / 1e6

    # ---- roofline of the dominant kernel (this rank's launch of render_kernel)
    plan_slices = ptamd.render_slices(cam, s_begin, s_end, n_split)
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    k_s = k_ms / 1000.0
    workload_name = f"{args.workload} {res[0]}x{res[1]} {spp}spp octree-depth-{meta['depth']}"
    # (a) demand on the scene arrays, by array: counters of this launch x bytes one unit of work needs.  These
    # are requests to the cache hierarchy (the geometry working set is L1 / L2 resident), not HBM traffic.
    hits = counters["segments"] - counters["env_lookups"] if counters["env_lookups"] else counters["segments"]
    demand = {"nodes": counters["node_visits"] * B_NODE, "triangles": counters["tri_tests"] * B_TRI, "shade_records": hits * B_SHADE,
              "texels": counters["tex_fetches"] * B_TEXEL, "hdri": counters["env_lookups"] * B_ENV,
              "entities": counters["segments"] * n_entities * B_ENTITY, "partials": n_pixels * 12 * plan_slices}
    alg_bytes = int(sum(demand.values()))
    fetch_demand = {k: round(v / k_s / 1e9, 1) for k, v in demand.items()} if k_ms > 0 else None
    # (b) compulsory HBM bytes of one launch: every scene array once, every slice partial written once
    sbytes = scene.device_bytes()
    partial_bytes = n_pixels * 12 * plan_slices
    compulsory = int(sum(sbytes.values()) + partial_bytes)
    # (c) measured memory-side traffic and VALU occupancy: separate rocprofv3 --pmc passes (they cannot run inside
    # this process), reported only when stamped with this workload and the current library sources
    tj = stamped_profile("traffic.json", workload_name, world)
    pj = stamped_profile("pmc_ratios.json", workload_name, world)
    traffic = fetch_b = write_b = None
    if tj:
        fetch_b, write_b = int(tj["FETCH_SIZE_KB"] * 1024 * 2), int(tj["WRITE_SIZE_KB"] * 1024)      # gfx950: FETCH_SIZE counts half
        traffic = fetch_b + write_b
    hbm = {"peak_GBps": HBM_PEAK_GBPS, "compulsory_bytes_per_launch": compulsory,
           "compulsory_frac": round(compulsory / k_s / 1e9 / HBM_PEAK_GBPS, 6) if k_ms > 0 else None,
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "traffic_frac": round(traffic / k_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic and k_ms > 0 else None,
           "write_amplification": round(write_b / partial_bytes, 2) if write_b else None,
           "traffic_source": tj["_file"] if tj else None}
    fj = stamped_profile("fetch_by_array.json", workload_name, world)
    if fj and k_ms > 0:
        # which array the memory-side fetches belong to (probe builds that keep one array in cache): the octree's nodes and
        # triangle records are in "geometry_and_scratch" together with the reloads of spilled registers
        hbm["fetch_GBps_by_array"] = {k: round(fj[k] / k_s / 1e9, 1) for k in ("textures", "hdri", "geometry_and_scratch")}
        hbm["fetch_by_array_source"] = fj["_file"]
    # (d) what binds: vector-instruction issue.  achieved = lane-operations of one launch (vector instructions x 64 x the mean
    # share of lanes active in one: separate --pmc passes over this very launch, a property of the work, not of the run) / the
    # kernel's duration measured live here; peak = every SIMD retiring 32 lanes per cycle at the nominal clock.
    n_mesh = meta["n_mesh"]
    alg_flops = algorithmic_flops(counters, n_mesh)
    tflops = alg_flops / k_s / 1e12 if k_ms > 0 else 0.0
    busy = pj.get("valu_busy") if pj else None
    lanes = pj.get("active_lanes_per_valu_instruction") if pj else None
    lane_ops = pj.get("valu_lane_ops_per_launch") if pj else None
    achieved = lane_ops / k_s / 1e12 if lane_ops and k_ms > 0 else None
    frac = round(achieved / LANE_OP_PEAK_T, 4) if achieved is not None else None
    roofline = {"bound": "valu-issue", "achieved": round(achieved, 2) if achieved is not None else None,
                "peak": round(LANE_OP_PEAK_T, 2), "unit": "Tlane-op/s", "frac": frac,
                "valu_instructions_per_launch": pj.get("valu_instructions_per_launch") if pj else None,
                "active_lane_fraction": lanes, "pmc_source": pj["_file"] if pj else None,
                # the instruction mix issues at about half rate (compares, selects on scalar masks, min / max, shifts, packed fp32:
                # profiles/r2/valu_issue_cost.txt); against a one-instruction-per-four-cycles ceiling the vector units are busy
                # `valu_busy_x4` of the time (raw counter ratio, may exceed 1: the mix also holds full-rate instructions)
                "half_rate_mix": {"valu_busy_x4": busy, "cycle_base": pj.get("cycle_base") if pj else None,
                                  "effective_clock_GHz": pj.get("effective_clock_GHz") if pj else None,
                                  "lane_slots_at_half_rate": round(busy * lanes, 4) if busy is not None and lanes is not None else None},
                "traffic": traffic, "kernel": "render_kernel", "kernel_ms": round(k_ms, 3),
                "hbm": hbm,
                "cache_demand": {"note": "requests to L1 / L2 by the in-kernel work counters x bytes per unit; NOT memory-side traffic (that is hbm.fetch / write_bytes_per_launch)",
                                 "bytes_per_launch": alg_bytes, "GBps": round(alg_bytes / k_s / 1e9, 1) if k_ms > 0 else None,
                                 "GBps_by_array": fetch_demand},
                "useful_fp32": {"TFLOPs": round(tflops, 3), "frac_of_vector_peak": round(tflops / VALU_PEAK_TFLOPS, 5),
                                "flops_per_launch": int(alg_flops)}}
    pmc = {k: v for k, v in pj.items() if k not in ("workload", "source_hash", "_file")} if pj else None

    cpu_baseline = None
    parity = None
    post_ms = None
    if rank == 0:
        # post-processing (reported separately): the reference's fixed chain, device resident
        out = torch.empty_like(local_sum)
        ptamd.check(L.pt_post_chain(dev, local_sum.data_ptr(), out.data_ptr(), res[0], res[1], spp, stream))   # loads the kernels
        torch.cuda.synchronize()
        tp = time.perf_counter()
        ptamd.check(L.pt_post_chain(dev, local_sum.data_ptr(), out.data_ptr(), res[0], res[1], spp, stream))
        torch.cuda.synchronize()
        post_ms = 1000.0 * (time.perf_counter() - tp)
        if args.save_ppm:
            sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
            from ptamd import io as ptio
            ptio.save_ppm(args.save_ppm, out.cpu().numpy())
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import oracle_api as O
            progress("CPU baseline: building the oracle's scene")
            osc, ocam = O.scene_from_description(desc)
            progress("CPU baseline: rendering")
            cores = min(os.cpu_count() or 1, 16)
            rows = (0, res[1])
            tc = time.perf_counter()
            osc.render(ocam, 1337, 0, 1, rows=rows, n_threads=cores)
            t1 = time.perf_counter() - tc
            n_cpu = int(max(1, min(256, round(15.0 / max(t1, 1e-3)))))
            tc = time.perf_counter()
            img_c, ccnt = osc.render(ocam, 1337, 1, 1 + n_cpu, rows=rows, n_threads=cores)
            tn = time.perf_counter() - tc
            # the same samples on the GPU: full-frame check of the HIP path against the oracle (untimed)
            # (with the counting build of the kernel, so that its work counters are checked as well and the timed
            # kernel's profile holds nothing but the warm-up and timed launches)
            img_g, gcnt = ptamd.render_image(scene, cam, n_cpu, seed=1337, sample_begin=1, device=dev, want_counters=True)
            diff = (img_g.astype(np.float64) - img_c.astype(np.float64)) / n_cpu
            rms = float(np.sqrt(np.mean((img_c.astype(np.float64) / n_cpu) ** 2)))
            ref_px = np.abs(img_c.astype(np.float64) / n_cpu).max(axis=-1)
            parity = {"rmse_over_rms": round(float(np.sqrt(np.mean(diff ** 2))) / max(rms, 1e-30), 6), "spp": n_cpu,
                      "pixels_differing": int((np.abs(diff).max(axis=-1) > 2e-4 * (rms + ref_px)).sum()), "pixels": n_pixels,
                      "counters_rel_diff": {k: round(abs(gcnt[k] - ccnt[k]) / max(ccnt[k], 1), 7)
                                            for k in ("paths", "segments", "node_visits", "tri_tests", "env_lookups", "tex_fetches")}}
            cpu_baseline = {"value": round(n_pixels * n_cpu / tn / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
                            "sample": f"full {res[0]}x{res[1]} frame, {n_cpu} spp, {cores} threads, {tn:.1f} s"}

    if rank == 0:
        line = {
            "metric": "Mpaths/s (+ time-to-2048spp), hcandersen 1080p" if args.workload == "hcandersen" else f"Mpaths/s, {args.workload}",
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name,
                       "triangles": meta["triangles"],
                       "spp": spp, "samples_per_rank": s_end - s_begin,
                       "parallelism": f"sample-shard x{world}" if tiles[1] == 1 else f"tile-shard x{world}"},
            "time_to_spp_s": round(ms_per_step / 1000.0, 4),
            "msegments_per_s": round(total_counters["segments"] / (ms_per_step / 1000.0) / 1e6, 2),
            "counters": total_counters, "schedule": schedule, "launch": launch,
            "roofline": roofline, "pmc": pmc, "cpu_baseline": cpu_baseline, "parity_vs_oracle": parity,
            "post_chain_ms": None if post_ms is None else round(post_ms, 2),
            "scene_gen_s": round(t_gen, 2), "scene_build_upload_s": round(t_upload, 2),
            "scene_by_rank": {"build_upload_s": [b[0] for b in build_info], "octree_builds": [b[1] for b in build_info]},
            "scene_device_bytes": sbytes, "library_source_hash": ptamd.source_hash(), "backend": args.backend if use_dist else None,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
