"""Throughput against the size of a step: samples per pixel per pt_trace call with the whole step in one pixel chunk
(pt_ctx_set_max_paths), one and two frames in flight; dragon stand-in at 1920x1080 (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pt_amd import hip
W, H = 1920, 1080
scene = bench.build_scene("dragon")
stream = torch.cuda.current_stream().cuda_stream
r = hip.Renderer(scene, W, H, device=0, stream=stream)
accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
for spp, mp, fif in [(64, 1 << 27, 1), (128, 1 << 28, 1), (192, 3 << 27, 1), (256, 1 << 29, 1), (384, 3 << 28, 1), (128, 1 << 28, 2), (192, 3 << 27, 2), (256, 1 << 29, 2)]:
    r.ctx.set_frames_in_flight(fif)
    r.ctx.set_max_paths(mp)
    for f in range(max(2, fif)):
        r.render(spp=spp, max_depth=10, accum_mode=hip.ACCUM_SUM_F32, accum_dev=accum.data_ptr(), frame=1000 + f)
    r.ctx.sync(); torch.cuda.synchronize(); r.reset_stats()
    t0 = time.perf_counter(); n = 4
    for f in range(n):
        r.render(spp=spp, max_depth=10, accum_mode=hip.ACCUM_SUM_F32, accum_dev=accum.data_ptr(), frame=f)
    r.ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = r.stats()
    print(f"spp {spp} max_paths {mp} fif {fif}: {1000*dt/n:.2f} ms/step, {int(st['rays'])/dt:.3e} rays/s", flush=True)
