"""world_size-2 gloo test of the multi-GPU path: frames sharded over ranks, one sum all-reduce.
The CPU oracle stands in for the per-rank renderer so the test runs without a GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = H = 24
SPP = 2
FRAMES = 5


def _render_sum(frames):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from pt_amd import camera
    from pt_amd.scene import Scene
    scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
    orc = O.OracleScene(scene)
    cam = camera.camera_uniform(W, H)
    acc = np.zeros((W * H, 4), np.float32)
    for f in frames:
        res, _ = orc.render_frame(cam, W, H, frame=f, spp=SPP, threads=1)
        O.accumulate(0, f, res, acc)
    return acc


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    from pt_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = torch.from_numpy(_render_sum(sharding.frames_for_rank(rank, world, FRAMES)))
    sharding.allreduce_sum(acc)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spp_sharding_matches_single_process(tmp_path):
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, 29517, out), nprocs=2, join=True)
    got = np.load(out)
    want = _render_sum(range(FRAMES))
    assert np.all(got[:, 3] == FRAMES)
    assert np.allclose(got[:, :3], want[:, :3], rtol=1e-6, atol=1e-7)   # float summation order differs


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pt_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_starts_its_own_ranks_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE: the parent starts torch.distributed.run with N ranks as a
    child process, hands on its own arguments and exit code, and imports neither torch nor the library."""
    import subprocess
    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("the parent must exit with the child's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_refuses_a_world_that_is_not_what_was_asked(monkeypatch):
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    try:
        bench.main()
        raise AssertionError("a line for 4 GPUs must not come from 2 ranks")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)
