#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary+secondary) and frame time of
scenes/chapter11.yaml at 10000x5000, recursion depth 30, on N GPUs of one node.

A step is one full frame: every rank renders its interleaved row blocks with
the HIP path (scene replicated, outputs in HBM) and the 8-bit row blocks are
gathered to rank 0 over RCCL.  `value` = (primary + secondary rays of the whole
frame) / frame time.  Total work is fixed as N grows (strong scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints one JSON line.  At N=1 it also carries `roofline` (algorithmic
f64 FLOP of the k_trace launches over their summed HIP-event time; the other
kernels are listed beside it) and `cpu_baseline` (the CPU oracle timed on a
bounded sample of rows of the same frame).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene file or None for --sphere-scene, scale, depth)
    "chapter11-10000x5000-depth30": ("scenes/chapter11.yaml", 10.0, 30),
    "chapter6-1000x1000": ("scenes/chapter6.yaml", 2.0, 5),
    "spheres-3000x2400-depth20": (None, 3.0, 20),
    "chapter15-dragon-1000x1000": ("scenes/chapter15-dragon.yaml", 1.0, 5),
    "bounding-box-1000x400": ("scenes/bounding-box.yaml", 1.0, 5),
    "chapter16-1000x1000": ("scenes/chapter16.yaml", 1.0, 5),
    "area-light-800x320": ("scenes/area-light.yaml", 1.0, 5),
}
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 FMA lanes x 2 flop x 2.4 GHz (half the 157.3 TF f32 vector rate)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="chapter11-10000x5000-depth30", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=None, help="override the workload's --scale")
    ap.add_argument("--depth", type=int, default=None, help="override the workload's --depth")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-rows", type=int, default=768, help="rows of the frame the CPU oracle renders for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-frames", action="store_true", help="skip the frames rendered into host memory after the timed region")
    ap.add_argument("--streams", type=int, default=2, help="image-row chunks in flight at once (1..4)")
    ap.add_argument("--grid-blocks", type=int, default=None)
    ap.add_argument("--traversal", type=int, default=None, help="0 auto, 1 wave-uniform, 2 per lane, 3 the reference walk (no device-only shortcut)")
    ap.add_argument("--chunk-pixels", type=int, default=None)
    ap.add_argument("--min-chunks", type=int, default=None, help="cut a frame of a megapixel or more into at least this many chunks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as ra
    from raytracer_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo between them - a rehearsal of the N>1 code path on a
    # box with one GPU (RCCL refuses two ranks on one device); never a measurement
    rehearsal = world > 1 and os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ra.init(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)

    scene_file, scale, depth = WORKLOADS[args.workload]
    if args.scale is not None:
        scale = args.scale
    if args.depth is not None:
        depth = args.depth
    if scene_file is None:
        scene = ra.Scene.generate_random_spheres(scale, args.seed)
    else:
        scene = ra.Scene.from_file(os.path.join(ROOT, scene_file), scale, args.seed)
    seeds = scene.row_seeds()
    scene.upload()
    scene.set_option(ra.OPT_STREAMS, args.streams)
    if args.grid_blocks:
        scene.set_option(ra.OPT_GRID_BLOCKS, args.grid_blocks)
    if args.traversal is not None:
        scene.set_option(ra.OPT_TRAVERSAL, args.traversal)
    if args.chunk_pixels:
        scene.set_option(ra.OPT_CHUNK_PIXELS, args.chunk_pixels)
    if args.min_chunks:
        scene.set_option(ra.OPT_MIN_CHUNKS, args.min_chunks)
    W, H = scene.width, scene.height
    part = multi.partition_for(rank, world) if world > 1 else None
    rows = len(ra.local_rows(H, part))
    dev = torch.device("cuda", local_rank)
    rgb = torch.empty((rows, W, 3), dtype=torch.float64, device=dev)
    rgb8 = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)

    def step():
        st = scene.render_device(depth, seeds, part, rgb.data_ptr(), rgb8.data_ptr())
        frame = multi.gather_rows(rgb8, H, rank, world)
        return st, frame

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    stats = None
    frame = None
    for _ in range(args.steps):
        stats, frame = step()
    fence()
    elapsed = time.perf_counter() - t0

    # whole-job numbers: max time over ranks, rays summed over ranks
    red_dev = torch.device("cpu") if rehearsal else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([stats["primary_rays"], stats["secondary_rays"], stats["shadow_rays"]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    primary, secondary, shadow = (int(v) for v in rays.tolist())
    ms_per_step = 1e3 * elapsed / args.steps
    mrays = (primary + secondary) / (ms_per_step * 1e-3) / 1e6

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+secondary), chapter11 10000x5000 depth 30"
            if args.workload.startswith("chapter11-10000") and args.scale is None and args.depth is None
            else "Mrays/s (primary+secondary)",
            "value": mrays,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "the reference's own scene file (%s, --seed %d); no synthetic substitute is needed" % (scene_file, args.seed)
            if scene_file else "generated sphere scene (--sphere-scene --seed %d)" % args.seed,
            "config": {
                "workload": args.workload, "width": W, "height": H, "depth": depth, "seed": args.seed,
                "parallelism": "rows in blocks of %d interleaved over %d GPU(s), scene replicated, u8 gather to rank 0"
                % (multi.BLOCK_ROWS, world),
            },
            "frame_ms": ms_per_step,
            **({"rehearsal": "all ranks on GPU 0, gloo between them: exercises the N>1 code path, measures nothing"} if rehearsal else {}),
            "rays": {"primary": primary, "secondary": secondary, "shadow": shadow,
                     "all_mrays_per_s": (primary + secondary + shadow) / (ms_per_step * 1e-3) / 1e6},
            "published_cpu_reference": {"rayon_s": 71.55, "single_thread_s": 264.26, "hardware": "unstated (README.md:120-126)"},
        }
        if rehearsal:
            # the assembled frame against one whole-frame render on this rank
            whole = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            scene.render_device(depth, seeds, None, 0, whole.data_ptr())
            torch.cuda.synchronize()
            out["rehearsal_frame_equals_single_render"] = bool(torch.equal(whole, frame))
        if world == 1:
            # Kernel times: with two chunks in flight the kernels of different chunks share the GPU and their
            # event spans overlap, so the per-kernel figures come from one extra frame rendered with a single
            # stream right after the timed region (same scene, same buffers, same kernels).
            timed_spans = {"streams": args.streams, "stream_ms_per_frame": {
                "k_trace": stats["ms_trace"], "k_shadow": stats["ms_shadow"], "k_shade": stats["ms_shade"]}}
            if args.streams != 1:
                scene.set_option(ra.OPT_STREAMS, 1)
                stats, _ = step()
                torch.cuda.synchronize()
                scene.set_option(ra.OPT_STREAMS, args.streams)
            fused = bool(stats["fused_shadows"])  # k_trace also casts the shadow rays (no area light)
            flop = stats["flop_trace"] + (stats["flop_shadow"] if fused else 0)
            ms_k = stats["ms_trace"]
            launches = stats["trace_launches"]
            dominant = "k_trace"
            if not fused and stats["ms_shadow"] > ms_k:
                # the shadow rays are a kernel of their own and take longer than the tracing (area lights: a hundred
                # shadow rays per shade point): the roofline is that kernel's, one launch per chunk
                dominant, flop, ms_k, launches = "k_shadow", stats["flop_shadow"], stats["ms_shadow"], stats["chunks"]
            achieved = flop / (ms_k * 1e-3) / 1e12 if ms_k > 0 else 0.0
            # algorithmic bytes of the record arrays (8-byte and 4-byte columns, one element per ray):
            #   every ray: hit_pos written once, read by k_shadow and k_shade (12 B)
            #   secondary ray: origin+direction written by the parent, read by k_trace (96 B);
            #                  colour written by k_shade, read by the parent (48 B)
            #   primary ray: direction written, read by k_shade (48 B); f64 + u8 frame buffers (27 B)
            #   shade point: over_point+normal written (48 B), over_point read by k_shadow (24 B),
            #                over_point+normal+direction read by k_shade (72 B, the direction's
            #                write is counted above), child links written + read (16 B), one
            #                intensity per light written + read (16 B)
            n_lights = len(scene.desc()["light_kind"])
            shade = stats["shade_hits"]
            alg_bytes = 12 * (primary + secondary) + 144 * secondary + 75 * primary + (48 + 24 + 72 + 16 + 16 * n_lights) * shade
            out["roofline"] = {
                "bound": "valu_fp64",
                "kernel": "k_trace (closest hit + n1/n2 + spawn" + (" + shadow rays" if fused else "") +
                          "; one launch per recursion level and chunk)" if dominant == "k_trace" else
                          "k_shadow (one closest-hit walk per light sample of every shade point; one launch per chunk)",
                "measured": "HIP events on the launch stream around each chunk's %s launches, in one extra "
                            "single-stream frame after the timed region" % dominant if args.streams != 1 else
                            "HIP events on the launch stream around each chunk's %s launches, last timed frame" % dominant,
                "single_stream_frame_ms": stats["ms_total"],
                "timed_region": timed_spans,
                "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VALU_PEAK_TFLOPS,
                "launches_per_frame": launches,
                "avg_launch_ms": ms_k / launches if launches else None,
                "algorithmic_flop_per_launch": flop / launches if launches else None,
                "algorithmic_flop_per_frame": flop,
                "kernel_ms_per_frame": ms_k,
                "traffic": None,
                "other_kernels": {
                    "k_trace": {"ms_per_frame": stats["ms_trace"], "algorithmic_flop_per_frame": stats["flop_trace"]},
                    "k_shadow": ({"fused_into": "k_trace", "algorithmic_flop_per_frame": stats["flop_shadow"]} if fused else
                                 {"ms_per_frame": stats["ms_shadow"], "algorithmic_flop_per_frame": stats["flop_shadow"]}),
                    "k_shade": {"ms_per_frame": stats["ms_shade"], "algorithmic_flop_per_frame": stats["flop_light"]},
                },
                "hbm": {"algorithmic_bytes_per_frame": alg_bytes,
                        "achieved_GBs_over_frame": alg_bytes / (ms_per_step * 1e-3) / 1e9,
                        "peak_GBs": HBM_PEAK_GBS},
            }
            if stats["mesh_nodes"]:
                # lane-walked meshes: what the walks fetch from the meshes' own hierarchies (128 B per node
                # visited, 80 B per triangle tested), over the time of the kernels that walk
                walk_ms = stats["ms_trace"] + stats["ms_shadow"]
                mesh_bytes = 128 * stats["mesh_nodes"] + 80 * stats["mesh_tris"]
                out["roofline"]["mesh_fetch"] = {
                    "nodes_per_frame": stats["mesh_nodes"], "triangles_per_frame": stats["mesh_tris"],
                    "nodes_per_ray": stats["mesh_nodes"] / max(1, primary + secondary + shadow),
                    "triangles_per_ray": stats["mesh_tris"] / max(1, primary + secondary + shadow),
                    "bytes_per_frame": mesh_bytes,
                    "achieved_GBs": mesh_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else None,
                    "note": "served by the L2 / Infinity Cache (the hierarchies are a few MB), so a latency figure, not an HBM one",
                }
            # k_trace's own share of the algorithmic bytes above: it reads a secondary ray's origin and direction (48 B)
            # and writes hit_pos (12 B per ray), each child's origin and direction (48 B), a primary ray's direction
            # (24 B), and per shade point over_point + normal (48 B), the child links (8 B) and one intensity per light
            alg_trace = 48 * secondary + 12 * (primary + secondary) + 48 * secondary + 24 * primary + \
                (48 + 8 + (8 * n_lights if fused else 0)) * shade
            out["roofline"]["traffic"], out["roofline"]["traffic_detail"] = measured_traffic(args, launches, alg_trace)
            if not args.no_host_frames:
                out.update(host_frames(ra, scene, depth, seeds, H, W, frame))
            if not args.no_cpu_baseline:
                out["cpu_baseline"], out["parity_on_sample"] = cpu_baseline(args, scene_file, scale, depth, frame, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, launches, alg_trace):
    """HBM-side bytes of k_trace per launch from the committed counter passes of this same command
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each; tools/collect_profiles.sh and
    tools/summarise_profiles.py).  Counters cannot be read from inside the process, so the line
    carries the newest committed profile of the headline workload and says which."""
    import glob
    if not (args.workload.startswith("chapter11-10000") and args.scale is None and args.depth is None) or not launches:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    k = t["kernels"].get("k_trace")
    if not k:
        return None, None
    per_launch = (k["fetch_bytes_corrected"] + k["write_bytes"]) / k["launches"]
    detail = dict(k)
    detail.update({"algorithmic_bytes_k_trace_per_frame": alg_trace,
                   "measured_over_algorithmic": (k["fetch_bytes_corrected"] + k["write_bytes"]) / alg_trace if alg_trace else None})
    detail.update({"source": os.path.relpath(files[-1], ROOT), "unit": "bytes per frame (k_trace launches summed)",
                   "correction": t.get("correction"), "other_kernels": {n: v for n, v in t["kernels"].items() if n != "k_trace"}})
    return per_launch, detail


def host_frames(ra, scene, depth, seeds, H, W, frame8_dev):
    """Frames delivered to host memory, as the reference's Camera::render returns its Canvas: wall time of
    rt_render with host outputs, outside the timed region.  `pinned`: the caller page-locked its buffer once
    (rt_host_register); `pageable`: an ordinary buffer, which rt_render locks for the length of each call."""
    out = {}
    rgb8 = np.empty((H, W, 3), dtype=np.uint8)
    rgb8[:] = 0   # touch every page before any timing
    rgb = np.empty((H, W, 3), dtype=np.float64)
    rgb[:] = 0

    def timed(a, b, reps=2):
        scene.render_into(depth, seeds, None, a, b)
        t0 = time.perf_counter()
        for _ in range(reps):
            scene.render_into(depth, seeds, None, a, b)
        return 1e3 * (time.perf_counter() - t0) / reps

    out["frame_ms_host_u8_pageable"] = timed(None, rgb8)
    out["frame_ms_host_f64_pageable"] = timed(rgb, None)
    same = bool((rgb8 == frame8_dev.cpu().numpy()).all())
    del rgb
    rgb8 = ra.host_empty((H, W, 3), np.uint8)
    rgb = ra.host_empty((H, W, 3), np.float64)
    out["frame_ms_host_u8"] = timed(None, rgb8)
    out["frame_ms_host_f64"] = timed(rgb, None)
    out["host_frames_note"] = ("rt_render into host memory, whole frame, mean of 2 after 1 warm-up; *_host_u8 / *_host_f64: the "
                               "caller's buffer is page-locked (rt_host_alloc), chunks arrive as they finish; *_pageable: "
                               "ordinary memory, filled through the library's page-locked staging buffer and a threaded memcpy")
    same = same and bool((rgb8 == frame8_dev.cpu().numpy()).all())
    out["host_u8_equals_device_u8"] = same
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Threads this process may keep busy: its affinity mask, cut to the cgroup's CPU quota and to the
    OMP_NUM_THREADS the box sets as the share of one GPU's job."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = "affinity mask"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < n:
                n, why = int(quota) // int(period), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and 0 < int(env) < n:
        n, why = int(env), "OMP_NUM_THREADS set by the box (one GPU's share of the host)"
    return n, why


def cpu_baseline(args, scene_file, scale, depth, frame8, height):
    """Time the CPU oracle (a port of the reference's algorithm, row-parallel with dynamic
    scheduling like its rayon path) on evenly spaced rows of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    cores, cores_rule = usable_cores()
    if scene_file is None:
        o = ol.OracleScene.spheres(scale, args.seed)
    else:
        o = ol.OracleScene.from_file(os.path.join(ROOT, scene_file), scale, args.seed)
    oseeds = o.row_seeds()
    n = max(1, min(args.cpu_rows, height))
    rows = np.unique(np.linspace(0, height - 1, n).astype(np.uint32))
    t0 = time.perf_counter()
    ref = o.render_row_list(depth, rows, oseeds, threads=cores)
    dt = time.perf_counter() - t0
    c = o.counters()
    rays = len(rows) * o.width + c["secondary"]
    ref8 = ol.to_u8(ref)
    got8 = frame8[torch_index(rows, frame8)].cpu().numpy()
    mism = int((got8 != ref8).any(axis=2).sum())
    base = {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "%d evenly spaced rows of the same %dx%d frame (%d primary + %d secondary rays) in %.2f s"
        % (len(rows), o.width, height, len(rows) * o.width, c["secondary"], dt),
        "projected_frame_s": dt * height / len(rows),
        "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "cores_rule": cores_rule,
    }
    parity = {"rows": int(len(rows)), "pixels": int(len(rows) * o.width), "u8_mismatch_pixels": mism}
    return base, parity


def torch_index(rows, like):
    import torch
    return torch.as_tensor(rows.astype(np.int64), device=like.device)


if __name__ == "__main__":
    main()
