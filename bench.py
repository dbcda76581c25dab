#!/usr/bin/env python3
"""Headline benchmark: rays/s and paths/s of the wavefront path tracer at 1080p.

    python bench.py --gpus N --steps K --warmup W

One step = one pt_trace call of --spp samples per pixel over the whole 1920x1080 image of the
dragon scene (procedural stand-in for the missing DragonAttenuation.bin), all inputs resident in
HBM.  The library can keep several steps in flight on internal streams; how many is chosen by
measurement during the warm-up unless --frames-in-flight says so.
With N > 1 (one process per GPU under torch.distributed.run) rank r renders frames
r, r+N, r+2N, ... (samples-per-pixel sharding) into its own float sum buffer and the buffers are
summed with one RCCL all-reduce inside the timed region.  By default every rank renders --steps
steps (per-GPU work fixed: weak scaling); with --total-spp S the S samples per pixel of ONE image
are split over the ranks (total work fixed: strong scaling, time to image).
Rank 0 prints one JSON line.  After the timed region, at one GPU, the metric's second scene (the
Cornell box) is timed for three steps and a bounded CPU sample of the workload is timed on the host.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak
CLOCK_GHZ = 2.4           # MI355X peak engine clock
VALU_PEAK_PER_SIMD = 0.5  # wave64 vector instructions per cycle per SIMD: 157.3 TFLOP/s fp32 = 256 CUs x 4 SIMDs x 64 lanes x 2 flop x 2.4 GHz x 0.5


def source_hash():
    """Identifies the sources of the tracing kernels and of the traversal layout a PMC record was collected with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "path-tracer_amd", "csrc")
    for name in ("pt_trace.hip", "pt_device.h", "pt_traverse_sched.h", "pt_internal.h", "pt_wide.h", "pt_sah.h", "pt_build.hip"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_record(workload_key):
    """Counters of the traversal launches of this workload from separate rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/pmc_traffic.py): memory-side bytes and wave64 vector
    instructions per launch.  None when no pass was recorded for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload_key)
    except (OSError, ValueError):
        return None


def make_roofline(launches, total_ms, rays, n_simds, pmc, compulsory_bytes, survey_bytes_per_ray, frames_in_flight, per_kernel):
    """The roofline object of the bench line for the traversal launches (k_primary + k_intersect), whose
    durations were measured with HIP events while each launch had the GPU to itself.

    The kernels are bound by vector-instruction issue, so that is the headline: achieved = wave64 vector
    instructions per launch (SQ_INSTS_VALU, PMC pass of the same workload and sources) / average launch
    duration; peak = 0.5 instructions per cycle per SIMD (what the 157.3 TFLOP/s fp32 figure implies)
    x SIMDs x 2.4 GHz.  `lane_utilisation` is the fraction of the 64 lanes that take part in an average vector
    instruction (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) and `useful_frac` = frac x lane_utilisation:
    the share of the peak spent on lanes that hold a ray.  Beside it `hbm`: measured memory-side bytes per launch
    / duration against 8 TB/s, and the bytes a launch cannot avoid (rays read, hit records and path state
    written).  `survey_model` keeps the modelled bytes per ray of a binary BVH in full precision (32 B per box
    test, 36 B per triangle test), which the quantised wide tree held in L2 does not move: it is not a bound."""
    avg_ms = total_ms / max(launches, 1)
    sec = avg_ms * 1e-3
    peak = VALU_PEAK_PER_SIMD * n_simds * CLOCK_GHZ      # G wave-instructions per second
    insts = float(pmc["valu_insts_per_launch"]) if pmc and pmc.get("valu_insts_per_launch") else None
    traffic = float(pmc["bytes_per_launch"]) if pmc and pmc.get("bytes_per_launch") else None
    lanes = float(pmc["lane_utilisation"]) if pmc and pmc.get("lane_utilisation") else None
    achieved = insts / sec / 1e9 if insts and sec > 0 else None
    frac = achieved / peak if achieved is not None else None
    # the same per kernel, where the PMC record holds the kernels apart
    per_kernel = {k: dict(v) for k, v in per_kernel.items()}
    for k, v in per_kernel.items():
        rec = (pmc or {}).get("kernels", {}).get(k)
        if rec and v.get("avg_launch_ms", 0) > 0:
            a = rec["valu_insts_per_launch"] / (v["avg_launch_ms"] * 1e-3) / 1e9
            v.update({"valu_insts_per_launch": rec["valu_insts_per_launch"], "frac": a / peak, "lane_utilisation": rec["lane_utilisation"],
                      "useful_frac": a / peak * rec["lane_utilisation"],
                      "hbm_frac": (rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]) / (v["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    out = {
        "bound": "valu", "kernel": "k_intersect + k_primary", "achieved": achieved, "peak": peak, "unit": "G wave64 vector instructions/s",
        "frac": frac, "traffic": traffic, "lane_utilisation": lanes, "useful_frac": frac * lanes if frac is not None and lanes is not None else None,
        "launches": launches, "avg_launch_ms": avg_ms, "rays_per_launch": rays / max(launches, 1), "frames_in_flight": frames_in_flight,
        "valu_insts_per_launch": insts, "valu_insts_per_ray": insts * launches / rays if insts and rays else None,
        "insts_per_cycle_per_simd": achieved / (n_simds * CLOCK_GHZ) if achieved is not None else None,
        "simds": n_simds, "clock_ghz": CLOCK_GHZ,
        "intersect_rays_per_s": rays / (total_ms * 1e-3) if total_ms > 0 else 0.0,
        "per_kernel": per_kernel,
        "pmc_source": None if not pmc else {"file": "profiles/traffic.json", "command": pmc.get("command"), "source_hash": pmc.get("source_hash"),
                                            "stale": pmc.get("source_hash") not in (None, source_hash())},
    }
    hbm_ach = traffic / sec / 1e9 if traffic and sec > 0 else None
    comp = compulsory_bytes / max(launches, 1)
    out["hbm"] = {"achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS if hbm_ach is not None else None,
                  "traffic": traffic, "compulsory_bytes_per_launch": comp,
                  "traffic_over_compulsory": traffic / comp if traffic and comp > 0 else None}
    if survey_bytes_per_ray:
        out["survey_model"] = {"bytes_per_ray": survey_bytes_per_ray,
                               "modelled_gbs": rays * survey_bytes_per_ray / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0,
                               "note": "box and triangle tests of the CPU port's binary BVH x 32 B / 36 B + 48 B per ray; a model, not traffic and not a bound"}
    return out


def roofline_from_profile(p, rays_per_depth, n_frames_stats, n_simds, wkey, survey, frames_in_flight):
    """make_roofline from a PtTraceProfile (event-timed traversal launches) and the rays per bounce of the same frames."""
    i_ms = float(p["intersect_ms"]) + float(p["primary_ms"])
    i_rays = float(p["intersect_rays"]) + float(p["primary_rays"])
    i_n = int(p["intersect_launches"]) + int(p["primary_launches"])
    per_kernel = {
        "k_primary": {"launches": int(p["primary_launches"]), "avg_launch_ms": float(p["primary_ms"]) / max(int(p["primary_launches"]), 1),
                      "rays_per_launch": float(p["primary_rays"]) / max(int(p["primary_launches"]), 1)},
        "k_intersect": {"launches": int(p["intersect_launches"]), "avg_launch_ms": float(p["intersect_ms"]) / max(int(p["intersect_launches"]), 1),
                        "rays_per_launch": float(p["intersect_rays"]) / max(int(p["intersect_launches"]), 1)}}
    # bytes a traversal launch cannot avoid: k_intersect reads a 32-byte ray and writes a 16-byte hit record per
    # ray; k_primary writes 64 bytes of path state per surviving path and a 16-byte result per path that ends
    frames_roof = max(int(p["primary_launches"]), 1)
    per_frame = [x / max(n_frames_stats, 1) for x in rays_per_depth]
    survivors = per_frame[1] if len(per_frame) > 1 else 0.0
    compulsory = frames_roof * (48.0 * sum(per_frame[1:]) + 64.0 * survivors + 16.0 * (per_frame[0] - survivors))
    return make_roofline(i_n, i_ms, i_rays, n_simds, pmc_record(wkey), compulsory, survey, frames_in_flight, per_kernel)


def build_scene(name):
    from pt_amd import procedural
    from pt_amd.scene import Scene
    if name == "dragon":
        return procedural.dragon_standin()
    if name == "composite":
        g = os.path.join(ROOT, "tests", "golden")
        return procedural.composite(Scene.load_npz(os.path.join(g, "scene_cornell.npz")),
                                    procedural.with_native_texture_sizes(Scene.load_npz(os.path.join(g, "scene_helmet.npz"))))
    scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
    if name == "helmet":
        scene = procedural.with_native_texture_sizes(scene)
    return scene


def scene_camera(name, W, H):
    from pt_amd import camera
    if name == "helmet":       # the reference's camera frames the 0.3-unit helmet tiny
        return camera.camera_uniform(W, H, eye=(0.15, 0.1, 0.75), center=(0.0, 0.0, 0.0), fov_deg=50.0)
    return camera.camera_uniform(W, H)


def host_threads():
    """Threads of the CPU baseline: the CPU share of this job (the harness exports it as OMP_NUM_THREADS;
    os.cpu_count() shows the whole machine, which a job sharing the host with others must not take)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS")
    return max(1, min(avail, int(share))) if share and share.isdigit() else max(1, avail)


def cpu_baseline(scene, cam, width, height, spp, max_depth):
    """Times the CPU oracle (a literal transcription of the shaders traversing its own BVH) on a
    bounded sample of the same workload: the full image at a reduced sample count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    threads = host_threads()
    orc = O.OracleScene(scene)
    t0 = time.perf_counter()
    _, st = orc.render_frame(cam, width, height, frame=0, spp=spp, max_depth=max_depth, threads=threads)
    dt = time.perf_counter() - t0
    rays = int(st["rays"])
    return {
        "value": rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
        "sample": f"{width}x{height}, {spp} spp, depth {max_depth}: {rays} rays in {dt:.2f} s (trace phase only) on {threads} threads, "
                  f"the job's CPU share of a host with {os.cpu_count()} logical CPUs",
        "host_logical_cpus": os.cpu_count(), "paths_per_s": int(st["paths"]) / dt,
        "box_tests_per_ray": int(st["box_tests"]) / rays, "tri_tests_per_ray": int(st["tri_tests"]) / rays,
        "hit_fraction": int(st["hits"]) / rays, "textured_hit_fraction": int(st["tex_hits"]) / rays,
    }


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: runs this same command line under torch.distributed.run with N
    ranks on this node (rendezvous on 127.0.0.1, a free port) as a child process; rank 0 of the child prints the
    JSON line on the inherited stdout.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL needs dmabuf IPC on these hosts
    print("bench.py: starting " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rank_identity(torch, local_rank, rehearsal):
    """What this rank runs on, for the line's `parallelism.ranks`: device index, name, PCI bus id."""
    p = torch.cuda.get_device_properties(local_rank)
    bus = None
    if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    return {"rank": int(os.environ.get("RANK", "0")), "device": local_rank, "name": p.name, "pci_bus_id": bus,
            "uuid": str(getattr(p, "uuid", "")) or None, "pid": os.getpid(), "rehearsal_shared_device": bool(rehearsal)}


def run_frames(renderer, accum_ptr, frames, spp, max_depth):
    from pt_amd import hip
    for f in frames:
        renderer.render(spp=spp, max_depth=max_depth, accum_mode=hip.ACCUM_SUM_F32, accum_dev=accum_ptr, frame=f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "cornell", "cube", "helmet", "composite"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256,
                    help="samples per pixel per step (one pt_trace call). The whole step is resident at once (128 bytes of queues "
                         "and records per path); every bounce level is one launch whose ramp-down is a fixed cost, so throughput rises "
                         "with the step. If the queues do not fit, the step is halved until they do")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="strong scaling: render ONE image of this many samples per pixel, its steps of --spp samples split over the ranks "
                         "(--steps is then derived: total-spp / spp / ranks, rounded up)")
    ap.add_argument("--max-paths", type=int, default=0,
                    help="paths per pixel chunk (library default 2^27); 0: the whole step in one chunk (at most 2^31 - 1 paths)")
    ap.add_argument("--max-depth", type=int, default=10)
    ap.add_argument("--max-leaf", type=int, default=0, help="largest BLAS leaf, in triangles (0: the library's default)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="steps the library keeps in flight on separate streams; 0 (default): one or two, whichever four timed "
                         "warm-up steps show to be faster on this machine (two must win by 1 %% to pay for the second set of queues)")
    ap.add_argument("--serial-roofline", action="store_true", help="accepted for older command lines")
    ap.add_argument("--no-serial-roofline", action="store_true",
                    help="with several frames in flight: skip the three steps with one frame in flight that follow the timed region at one GPU "
                         "(their launch durations, each launch alone on the GPU, are what `roofline` is computed from; without them it is "
                         "computed from the timed region's overlapping launches and says so)")
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--share-primary", action="store_true",
                    help="trace bounce 0 once per pixel and share its hit among the pixel's samples (pt_ctx_set_share_primary: identical "
                         "images, fewer rays traversed).  Off by default: the headline counts one traceRayEXT per sample and bounce, as the "
                         "reference executes them; the default line reports the option's effect under `share_primary`")
    ap.add_argument("--secondary", action="store_true", help="accepted for older command lines; the Cornell measurement is on by default")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Cornell steps after the timed region (profiled runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start one rank per GPU as a child process and relay its line and exit code.
        # This process has not touched the GPU (torch is not even imported yet) and never will.
        raise SystemExit(launch_ranks(args.gpus))

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
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe what ran")
    visible = torch.cuda.device_count()
    if not rehearsal and local_rank >= visible:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but {visible} device(s) are visible")
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    identity = rank_identity(torch, local_rank, rehearsal)
    ranks_seen = [identity]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, identity)
        devices = {(r["pci_bus_id"], r["device"]) for r in ranks_seen}
        if not rehearsal and len(devices) != world:
            raise SystemExit(f"bench.py: {world} ranks share {len(devices)} device(s): {sorted(devices)}")
    try:
        rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        rccl_version = None

    trace_t0 = time.perf_counter()

    def phase(what):          # PT_BENCH_TRACE=1: where the wall-clock goes, per rank, on stderr
        if os.environ.get("PT_BENCH_TRACE") == "1":
            print(f"[bench rank {rank}] {time.perf_counter() - trace_t0:8.3f} s  {what}", file=sys.stderr, flush=True)

    W, H = args.width, args.height
    scene = build_scene(args.scene)
    cam = scene_camera(args.scene, W, H)
    # a stream of our own, made torch's current one: the library enqueues on it (its handle is not the null
    # stream's 0, which pt_ctx_create would take as "create one"), so torch's work and the library's are ordered
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0
    t0 = time.perf_counter()
    r = hip.Renderer(scene, W, H, device=local_rank, stream=stream, camera=cam, max_leaf=args.max_leaf)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    r.ctx.set_share_primary(args.share_primary)
    phase("renderer built")
    build_ms = sum(float(b.info()["build_ms"]) for b in r.blases) + float(r.tlas.info()["build_ms"])
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def set_chunk():
        r.ctx.set_max_paths(args.max_paths if args.max_paths > 0 else min(W * H * args.spp, (1 << 31) - 64))

    def my_frames(n_steps_per_rank, first):
        return sharding.frames_for_rank(rank, world, n_steps_per_rank * world, first)

    # the first steps allocate the queues of every frame in flight: a step that does not fit the
    # device's memory is halved (same work per ray, more launches per sample)
    fif = args.frames_in_flight if args.frames_in_flight > 0 else 1
    r.ctx.set_frames_in_flight(fif)
    while True:
        set_chunk()
        try:
            run_frames(r, accum.data_ptr(), my_frames(fif, 2_000_000), args.spp, args.max_depth)
            r.ctx.sync()
            break
        except hip.PtError as e:
            if "memory" not in str(e).lower() or args.spp <= 16:
                raise
            print(f"bench.py: {args.spp} spp per step does not fit ({e}); halving the step", file=sys.stderr)
            args.spp //= 2
    phase(f"first step done ({args.spp} spp)")
    fif_choice = None
    if args.frames_in_flight == 0:
        # one or two steps in flight, by measurement: two cost a second set of queues and must earn it
        def timed(n_fif):
            r.ctx.set_frames_in_flight(n_fif)
            run_frames(r, accum.data_ptr(), my_frames(n_fif, 2_100_000), args.spp, args.max_depth)   # allocates this setting's queues
            r.ctx.sync()
            t = time.perf_counter()
            run_frames(r, accum.data_ptr(), my_frames(4, 2_200_000), args.spp, args.max_depth)   # two steps overstate the overlap
            r.ctx.sync()
            return (time.perf_counter() - t) / 4
        t1 = timed(1)
        try:
            t2 = timed(2)
        except hip.PtError as e:
            if "memory" not in str(e).lower():
                raise
            t2 = float("inf")
        fif = 2 if t2 < 0.99 * t1 else 1
        fif_choice = {"ms_per_step_1": 1e3 * t1, "ms_per_step_2": None if t2 == float("inf") else 1e3 * t2, "chosen": fif}
        if world > 1:       # every rank runs the same setting
            tf = torch.tensor([fif], dtype=torch.int64, device="cuda")
            dist.all_reduce(tf, op=dist.ReduceOp.MIN)
            fif = int(tf[0])
        r.ctx.set_frames_in_flight(fif)
    if world > 1:
        # every rank runs the same step
        t_spp = torch.tensor([args.spp], dtype=torch.int64, device="cuda")
        dist.all_reduce(t_spp, op=dist.ReduceOp.MIN)
        args.spp = int(t_spp[0])
        set_chunk()
    phase(f"step agreed: {args.spp} spp, {fif} in flight")
    strong = args.total_spp > 0
    if strong:
        total_steps = max(1, -(-args.total_spp // args.spp))
        my_timed = sharding.frames_for_rank(rank, world, total_steps, 0)     # ranks may differ by one step
        args.steps = -(-total_steps // world)
    else:
        total_steps = args.steps * world
        my_timed = my_frames(args.steps, 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (frames that are not part of the timed set); at least one step per frame in flight, so
    # that every internal stream has allocated its path queues before the clock starts
    run_frames(r, accum.data_ptr(), my_frames(max(args.warmup, fif), 1_000_000), args.spp, args.max_depth)
    r.ctx.sync()                    # frames in flight still add to the image
    phase("warm-up steps done")
    sharding.allreduce_sum(accum)   # the collective's first call sets up its communicator and buffers: not timed
    phase("warm-up all-reduce done")
    accum.zero_()
    r.reset_stats()
    # The traversal launches are bracketed by HIP events on the streams they run on: during the timed steps when
    # one step runs at a time; with several in flight the event records between the launches cost 3-4 % of the
    # overlap, and the durations that count are those of the steps after the timed region anyway.
    serial_after = fif > 1 and rank == 0 and not args.no_serial_roofline
    r.set_profiling(not serial_after)
    r.profile()
    barrier()
    t0 = time.perf_counter()
    run_frames(r, accum.data_ptr(), my_timed, args.spp, args.max_depth)
    phase("timed steps enqueued")
    # The host waits for the frames in flight before the collective is issued (some 20 us on a step of 120 ms).
    # A stream-side wait (r.fence()) would do for RCCL, but the gloo rehearsal, whose all-reduce stages through
    # pinned host memory from a thread of its own, then crawled with two ranks on one GPU: launches of 10 s and more.
    r.ctx.sync()
    phase("timed steps done")
    ar0, ar1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ar0.record()
    sharding.allreduce_sum(accum)
    ar1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    phase("timed region closed")
    allreduce_ms = ar0.elapsed_time(ar1) if world > 1 else 0.0
    prof = r.profile()
    st = r.stats()
    rays_local, paths_local = int(st["rays"]), int(st["paths"])
    rays_per_depth = [int(x) for x in st["rays_per_depth"][:args.max_depth]]
    t = torch.tensor([elapsed, float(rays_local), float(paths_local)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays, paths = float(t[1]), float(t[2])
    else:
        rays, paths = float(rays_local), float(paths_local)
    # the reduced buffer holds every frame of every rank
    frames_seen = float(accum.view(-1, 4)[0, 3])
    if frames_seen != float(total_steps):
        raise SystemExit(f"bench.py: the reduced image holds {frames_seen} frames, {total_steps} were rendered by {world} rank(s)")
    # the same kernels with one frame at a time, so that a launch has the GPU to itself (not part of `value`)
    prof_roof, roof_fif = prof, fif
    if serial_after:
        r.ctx.set_frames_in_flight(1)
        r.set_profiling(True)
        run_frames(r, accum.data_ptr(), [3_000_000], args.spp, args.max_depth)
        r.profile()
        run_frames(r, accum.data_ptr(), [3_000_001, 3_000_002], args.spp, args.max_depth)
        prof_roof, roof_fif = r.profile(), 1
    r.set_profiling(False)

    if rank == 0:
        out = {
            "metric": "rays/s", "value": rays / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {args.spp} spp per step, max depth {args.max_depth}"
                                   + (" (procedural stand-in for the missing DragonAttenuation.bin: 134,995 triangles, 2 instances)" if args.scene == "dragon" else ""),
                       "parallelism": {"scheme": f"spp-sharded x{world}, one sum all-reduce of the float image" if world > 1 else "single GPU",
                                       "backend": backend, "world_size": world, "devices_visible": visible, "rccl_version": rccl_version,
                                       "ranks": ranks_seen, "allreduce_ms": allreduce_ms,
                                       "allreduce_bytes": accum.numel() * 4 if world > 1 else 0,
                                       "steps_total": total_steps, "total_spp": total_steps * args.spp},
                       "frames_in_flight": fif, "triangles": scene.triangle_count(), "instances": len(scene.instances()),
                       "share_primary": bool(args.share_primary)},
            "paths_per_s": paths / elapsed, "rays_per_path": rays / paths, "time_to_image_s": elapsed,
            "frames_accumulated": frames_seen, "bvh_build_ms": build_ms, "setup_s": setup_s,
        }
        if fif_choice:
            out["config"]["frames_in_flight_choice"] = fif_choice
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene, cam, W, H, args.cpu_spp, args.max_depth)
            out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "host_logical_cpus", "paths_per_s")}
        survey = (32.0 * cpu["box_tests_per_ray"] + 36.0 * cpu["tri_tests_per_ray"] + 48.0) if cpu else None
        wkey = f"{args.scene} {W}x{H} {args.spp}spp depth{args.max_depth}"
        n_simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
        out["roofline"] = roofline_from_profile(prof_roof, rays_per_depth, len(my_timed), n_simds, wkey, survey, roof_fif)
        out["roofline"]["durations_from"] = ("the timed region" if prof_roof is prof else "three steps after the timed region") + \
                                            f", {roof_fif} frame(s) in flight" + ("" if roof_fif == 1 else ": launches overlap, durations are stretched")
        if world == 1 and not args.no_secondary and not args.share_primary:
            # the same steps with bounce 0 traversed once per pixel (same images; fewer rays, so paths/s is the figure to compare)
            r.ctx.set_frames_in_flight(fif)
            r.ctx.set_share_primary(True)
            run_frames(r, accum.data_ptr(), range(4_000_000, 4_000_000 + max(2, fif)), args.spp, args.max_depth)
            r.ctx.sync()
            r.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_frames(r, accum.data_ptr(), range(4_100_000, 4_100_003), args.spp, args.max_depth)
            r.ctx.sync()
            dt = time.perf_counter() - t1
            s3 = r.stats()
            out["share_primary"] = {"ms_per_step": 1e3 * dt / 3, "steps": 3, "paths_per_s": int(s3["paths"]) / dt, "rays_traversed_per_s": int(s3["rays"]) / dt,
                                    "time_to_image_s_at_equal_spp": dt / 3 * args.steps, "speedup_in_paths_per_s": (int(s3["paths"]) / dt) / (paths / elapsed),
                                    "note": "pt_ctx_set_share_primary(1): bounce 0 traversed once per pixel, images bit-identical; not part of `value`"}
            r.ctx.set_share_primary(False)
        if world == 1 and not args.no_secondary and args.scene == "dragon":
            # the metric's other scene, after the timed region
            r.close()
            c = hip.Renderer(build_scene("cornell"), W, H, device=local_rank, stream=stream)
            c.ctx.set_frames_in_flight(fif)
            c.ctx.set_max_paths(args.max_paths if args.max_paths > 0 else min(W * H * args.spp, (1 << 31) - 64))
            acc2 = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
            run_frames(c, acc2.data_ptr(), range(1_000_000, 1_000_000 + max(2, fif)), args.spp, args.max_depth)
            c.ctx.sync()
            c.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_frames(c, acc2.data_ptr(), range(3), args.spp, args.max_depth)
            c.ctx.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            s2 = c.stats()
            out["cornell"] = {"rays_per_s": int(s2["rays"]) / dt, "paths_per_s": int(s2["paths"]) / dt, "ms_per_step": 1e3 * dt / 3, "steps": 3,
                              "workload": f"cornell {W}x{H}, {args.spp} spp per step, max depth {args.max_depth}, after the timed region"}
            # its own roofline: two more steps, one at a time, every traversal launch between HIP events
            c.ctx.set_frames_in_flight(1)
            c.reset_stats()
            c.set_profiling(True)
            c.profile()
            run_frames(c, acc2.data_ptr(), range(10, 12), args.spp, args.max_depth)
            pc = c.profile()
            c.set_profiling(False)
            s2 = c.stats()
            out["cornell"]["roofline"] = roofline_from_profile(pc, [int(x) for x in s2["rays_per_depth"][:args.max_depth]], 2, n_simds,
                                                               f"cornell {W}x{H} {args.spp}spp depth{args.max_depth}", None, 1)
            c.close()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
