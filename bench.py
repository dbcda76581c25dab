#!/usr/bin/env python3
"""Headline benchmark: rays/s and paths/s of the wavefront path tracer at 1080p.

    python bench.py --gpus N --steps K --warmup W

One step = one pt_trace call of --spp samples per pixel over the whole 1920x1080 image of the
dragon scene (procedural stand-in for the missing DragonAttenuation.bin), all inputs resident in
HBM; --frames-in-flight steps may overlap on the library's internal streams (default here: 2, the
library's own default is 3).
With N > 1 (one process per GPU under torch.distributed.run) rank r renders frames
r, r+N, r+2N, ... (samples-per-pixel sharding) into its own float sum buffer and the buffers are
summed with one RCCL all-reduce inside the timed region; per-GPU work is fixed, so scaling is weak.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def measured_traffic(workload_key):
    """HBM-side bytes per intersect launch from the committed PMC passes (profiles/traffic.json,
    written by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this workload);
    None when no pass was recorded for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(workload_key)
        return float(rec["bytes_per_launch"]) if rec else None
    except (OSError, ValueError, KeyError):
        return None


def valu_issue(workload_key, avg_launch_ms, n_simds):
    """Vector-instruction issue rate of the traversal launches: wave64 vector instructions per launch
    (SQ_INSTS_VALU of this workload, profiles/traffic.json) over the launches' average duration measured
    in this run, per SIMD and 2.4 GHz cycle.  The ceilings are what one SIMD issues of a single
    instruction kind with 2-8 waves (tools/micro/valu_rate.hip): 0.40 for the full-rate kind (fma, add, mul,
    and/or/xor, mov), 0.235 for the half-rate kind (cndmask, min/max, compares, conversions), of which
    a traversal step is two thirds.  With several frames in flight the launches share the GPU and
    the rate per launch is lower by the stretch."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(workload_key)
        n = float(rec["valu_insts_per_launch"])
    except (OSError, ValueError, KeyError, TypeError):
        return None
    if avg_launch_ms <= 0:
        return None
    return {"insts_per_cycle_per_simd": n / (avg_launch_ms * 1e-3 * n_simds * 2.4e9), "valu_insts_per_launch": n,
            "simds": n_simds, "clock_ghz": 2.4, "ceiling_half_rate_kind": 0.235, "ceiling_full_rate_kind": 0.40}


def build_scene(name):
    from pt_amd import procedural
    from pt_amd.scene import Scene
    if name == "dragon":
        return procedural.dragon_standin()
    return Scene.load_npz(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))


def cpu_baseline(scene, width, height, spp, max_depth):
    """Times the CPU oracle (a literal transcription of the shaders traversing its own BVH) on a
    bounded sample of the same workload: the full image at a reduced sample count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from pt_amd import camera
    threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    orc = O.OracleScene(scene)
    cam = camera.camera_uniform(width, height)
    t0 = time.perf_counter()
    _, st = orc.render_frame(cam, width, height, frame=0, spp=spp, max_depth=max_depth, threads=threads)
    dt = time.perf_counter() - t0
    rays = int(st["rays"])
    return {
        "value": rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
        "sample": f"{width}x{height}, {spp} spp, depth {max_depth}: {rays} rays in {dt:.2f} s (trace phase only)",
        "paths_per_s": int(st["paths"]) / dt,
        "box_tests_per_ray": int(st["box_tests"]) / rays, "tri_tests_per_ray": int(st["tri_tests"]) / rays,
        "hit_fraction": int(st["hits"]) / rays, "textured_hit_fraction": int(st["tex_hits"]) / rays,
    }


def run_steps(renderer, accum_ptr, n_steps, first_frame, rank, world, spp, max_depth):
    """Rank `rank` renders its share of the frames [first_frame, first_frame + n_steps * world)."""
    from pt_amd import hip, sharding
    for f in sharding.frames_for_rank(rank, world, n_steps * world, first_frame):
        renderer.render(spp=spp, max_depth=max_depth, accum_mode=hip.ACCUM_SUM_F32, accum_dev=accum_ptr, frame=f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "cornell", "cube"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256,
                    help="samples per pixel per step (one pt_trace call). The whole step is resident at once (160 bytes of queues "
                         "and records per path: 85 GB per step in flight at 1080p x 256); throughput rises with the step because "
                         "every bounce level is one launch whose ramp-down is a fixed cost: +4 %% from 64 to 128 spp, +6 %% to 256 on the "
                         "dragon frame. If the queues do not fit, the step is halved until they do")
    ap.add_argument("--max-paths", type=int, default=0,
                    help="paths per pixel chunk (library default 2^27); 0: the whole step in one chunk")
    ap.add_argument("--max-depth", type=int, default=10)
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="steps the library keeps in flight on separate streams (library default 3). With two, the shading and "
                         "bounce-0 kernels of one step overlap the traversal of the other: +5.6 %% at 64 spp per step. Launches "
                         "that share the GPU last longer than they would alone: --frames-in-flight 1 (or --serial-roofline) gives "
                         "the kernels' exclusive durations")
    ap.add_argument("--serial-roofline", action="store_true", help="accepted for older command lines; on by default at one GPU")
    ap.add_argument("--no-serial-roofline", action="store_true",
                    help="skip the three steps with one frame in flight that follow the timed region at one GPU (their kernel durations, "
                         "each launch alone on the GPU, are reported as roofline.one_frame_at_a_time and are not part of `value`); "
                         "for profiled runs, whose kernel statistics should hold the timed command's launches only")
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true", help="also measure the cornell scene (extra launches: keep it out of profiled runs)")
    ap.add_argument("--no-secondary", action="store_true", help="accepted for older command lines; the side measurement is off by default")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pt_amd import hip, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a box with one GPU: every rank on device 0, gloo instead of RCCL
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H = args.width, args.height
    scene = build_scene(args.scene)
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    r = hip.Renderer(scene, W, H, device=local_rank, stream=stream)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    build_ms = sum(float(b.info()["build_ms"]) for b in r.blases) + float(r.tlas.info()["build_ms"])
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
    r.ctx.set_frames_in_flight(args.frames_in_flight)
    # the first steps allocate the queues of every frame in flight: a step that does not fit the
    # device's memory is halved (same work per ray, more launches per sample)
    while True:
        r.ctx.set_max_paths(args.max_paths if args.max_paths > 0 else min(W * H * args.spp, (1 << 31) - 64))
        try:
            run_steps(r, accum.data_ptr(), args.frames_in_flight, 2_000_000, rank, world, args.spp, args.max_depth)
            r.ctx.sync()
            break
        except hip.PtError as e:
            if "memory" not in str(e).lower() or args.spp <= 16:
                raise
            print(f"bench.py: {args.spp} spp per step does not fit ({e}); halving the step", file=sys.stderr)
            args.spp //= 2
    if world > 1:
        # every rank runs the same step
        t_spp = torch.tensor([args.spp], dtype=torch.int64, device="cuda")
        dist.all_reduce(t_spp, op=dist.ReduceOp.MIN)
        args.spp = int(t_spp[0])
        r.ctx.set_max_paths(args.max_paths if args.max_paths > 0 else min(W * H * args.spp, (1 << 31) - 64))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (frames that are not part of the timed set); at least one step per frame in flight, so
    # that every internal stream has allocated its path queues before the clock starts
    run_steps(r, accum.data_ptr(), max(args.warmup, args.frames_in_flight), 1_000_000, rank, world, args.spp, args.max_depth)
    r.ctx.sync()                    # frames in flight still add to the image
    sharding.allreduce_sum(accum)   # the collective's first call sets up its communicator and buffers: not timed
    accum.zero_()
    r.reset_stats()
    # the dominant kernel (intersect) is bracketed by HIP events on the streams it runs on, during the timed steps
    r.set_profiling(True)
    r.profile()
    barrier()
    t0 = time.perf_counter()
    run_steps(r, accum.data_ptr(), args.steps, 0, rank, world, args.spp, args.max_depth)
    r.ctx.sync()                    # the all-reduce (torch's stream) follows every frame in flight
    sharding.allreduce_sum(accum)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = r.profile()
    st = r.stats()
    rays_local, paths_local = int(st["rays"]), int(st["paths"])
    t = torch.tensor([elapsed, float(rays_local), float(paths_local)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays, paths = float(t[1]), float(t[2])
    else:
        rays, paths = float(rays_local), float(paths_local)
    # sanity: the reduced buffer holds every frame of every rank
    frames_seen = float(accum.view(-1, 4)[0, 3])
    # the same kernel with one frame at a time, so that a launch has the GPU to itself (not part of `value`)
    prof_serial = None
    if rank == 0 and world == 1 and args.frames_in_flight > 1 and not args.no_serial_roofline:
        r.ctx.set_frames_in_flight(1)
        run_steps(r, accum.data_ptr(), 1, 3_000_000, 0, 1, args.spp, args.max_depth)
        r.profile()
        run_steps(r, accum.data_ptr(), 2, 3_000_001, 0, 1, args.spp, args.max_depth)
        prof_serial = r.profile()
    r.set_profiling(False)

    if rank == 0:
        out = {
            "metric": "rays/s", "value": rays / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {args.spp} spp per step, max depth {args.max_depth}"
                                   + (" (procedural stand-in for the missing DragonAttenuation.bin: 134,995 triangles, 2 instances)" if args.scene == "dragon" else ""),
                       "parallelism": f"spp-sharded x{world}, one RCCL all-reduce of the float sum buffer" if world > 1 else "single GPU",
                       "triangles": scene.triangle_count(), "instances": len(scene.instances())},
            "paths_per_s": paths / elapsed, "rays_per_path": rays / paths,
            "frames_accumulated": frames_seen, "bvh_build_ms": build_ms, "setup_s": setup_s,
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene, W, H, args.cpu_spp, args.max_depth)
            out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample")}
            out["cpu_baseline"]["paths_per_s"] = cpu["paths_per_s"]
        # algorithmic bytes of one intersect launch = rays x (32 B per box test + 36 B per triangle
        # test + 32 B ray read + 16 B hit record written); box/triangle tests per ray are the CPU
        # oracle's counts on its canonical BVH (fallback: the figures measured for this scene).
        vn = cpu["box_tests_per_ray"] if cpu else 52.0
        vt = cpu["tri_tests_per_ray"] if cpu else 2.7
        hitf = cpu["hit_fraction"] if cpu else 0.46
        texf = cpu["textured_hit_fraction"] if cpu else 0.29
        bytes_intersect = 32.0 * vn + 36.0 * vt + 48.0
        bytes_pipeline = 32.0 * vn + 36.0 * vt + hitf * 336.0 + texf * 16.0 + 176.0
        # the traversal launches of a step: k_primary (bounce 0; its interval also holds the bounce's
        # shading, which makes the figure conservative) and k_intersect (one per later bounce)
        i_ms = float(prof["intersect_ms"]) + float(prof["primary_ms"])
        i_rays = float(prof["intersect_rays"]) + float(prof["primary_rays"])
        i_n = int(prof["intersect_launches"]) + int(prof["primary_launches"])
        achieved = (i_rays * bytes_intersect) / (i_ms * 1e-3) / 1e9 if i_ms > 0 else 0.0
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_intersect + k_primary", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(f"{args.scene} {W}x{H} {args.spp}spp depth{args.max_depth}"),
            "launches": i_n, "avg_launch_ms": i_ms / max(i_n, 1), "rays_per_launch": i_rays / max(i_n, 1),
            "frames_in_flight": args.frames_in_flight,
            "bytes_per_ray": bytes_intersect, "box_tests_per_ray": vn, "tri_tests_per_ray": vt,
            "intersect_rays_per_s": i_rays / (i_ms * 1e-3) if i_ms > 0 else 0.0,
            "per_kernel": {
                "k_primary": {"launches": int(prof["primary_launches"]),
                              "avg_launch_ms": float(prof["primary_ms"]) / max(int(prof["primary_launches"]), 1),
                              "rays_per_launch": float(prof["primary_rays"]) / max(int(prof["primary_launches"]), 1)},
                "k_intersect": {"launches": int(prof["intersect_launches"]),
                                "avg_launch_ms": float(prof["intersect_ms"]) / max(int(prof["intersect_launches"]), 1),
                                "rays_per_launch": float(prof["intersect_rays"]) / max(int(prof["intersect_launches"]), 1)}},
            "pipeline_bytes_per_ray": bytes_pipeline,
            "pipeline_frac": (rays / elapsed / world) * bytes_pipeline / 1e9 / HBM_PEAK_GBS,
        }
        wkey = f"{args.scene} {W}x{H} {args.spp}spp depth{args.max_depth}"
        n_simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
        vi = valu_issue(wkey, i_ms / max(i_n, 1), n_simds)
        if vi is not None:
            out["roofline"]["valu_issue"] = vi
        if prof_serial is not None and float(prof_serial["intersect_ms"]) > 0:
            s_ms = float(prof_serial["intersect_ms"]) + float(prof_serial["primary_ms"])
            s_rays = float(prof_serial["intersect_rays"]) + float(prof_serial["primary_rays"])
            s_n = int(prof_serial["intersect_launches"]) + int(prof_serial["primary_launches"])
            s_ach = (s_rays * bytes_intersect) / (s_ms * 1e-3) / 1e9
            # launches of frames in flight share the GPU, which stretches each of them; this is the
            # same measurement with frames strictly one after the other
            out["roofline"]["one_frame_at_a_time"] = {"achieved": s_ach, "frac": s_ach / HBM_PEAK_GBS, "launches": s_n,
                                                      "avg_launch_ms": s_ms / max(s_n, 1), "intersect_rays_per_s": s_rays / (s_ms * 1e-3)}
            vs = valu_issue(wkey, s_ms / max(s_n, 1), n_simds)
            if vs is not None:
                out["roofline"]["one_frame_at_a_time"]["valu_issue"] = vs
        if world == 1 and args.secondary and not args.no_secondary and args.scene == "dragon":
            r.close()
            c = hip.Renderer(build_scene("cornell"), W, H, device=local_rank, stream=stream)
            acc2 = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
            run_steps(c, acc2.data_ptr(), 3, 1_000_000, 0, 1, args.spp, args.max_depth)
            c.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(c, acc2.data_ptr(), 3, 0, 0, 1, args.spp, args.max_depth)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            s2 = c.stats()
            out["cornell"] = {"rays_per_s": int(s2["rays"]) / dt, "paths_per_s": int(s2["paths"]) / dt,
                              "workload": f"cornell {W}x{H}, {args.spp} spp per step, 3 steps"}
            c.close()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
