"""The C++ host mirror (Renderer / ModelLoader / BLAS / TLAS / BindlessResources in
path-tracer_amd/host) driven through its headless executable, checked against the CPU oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PT_RENDER = os.path.join(ROOT, "path-tracer_amd", "lib", "pt_render")


def _run(args):
    out = subprocess.run([PT_RENDER] + args, check=True, capture_output=True, text=True, timeout=110)
    return json.loads(out.stdout.strip().splitlines()[-1])


def _oracle_frame(scene, w, h, spp, frame=0, **cam):
    import oracle as O
    from pt_amd import camera
    return O.OracleScene(scene).render_frame(camera.camera_uniform(w, h, **cam), w, h, frame=frame, spp=spp, threads=8)


@pytest.mark.gpu
def test_cpp_renderer_ptscene_bit_exact(tmp_path, cornell):
    p = str(tmp_path / "cornell.ptscene")
    cornell.save_ptscene(p)
    dump = str(tmp_path / "frame.f32")
    w, h, spp = 96, 80, 8
    stats = _run(["--scene", p, "--width", str(w), "--height", str(h), "--spp", str(spp), "--frames", "1", "--dump-result", dump,
                  "--output", str(tmp_path / "img.ppm")])
    got = np.fromfile(dump, dtype=np.float32).reshape(h, w, 3)
    want, st = _oracle_frame(cornell, w, h, spp)
    assert np.array_equal(got, want)
    assert stats["rays"] == int(st["rays"]) and stats["paths"] == w * h * spp
    assert os.path.getsize(str(tmp_path / "img.ppm")) > w * h * 3


@pytest.mark.gpu
def test_cpp_gltf_and_png_import_bit_exact(tmp_path, cube):
    """glTF parsing, PNG decoding and the bindless texture path of the C++ host: the cube fixture is
    written out as .gltf + .png, imported by the C++ ModelLoader and rendered."""
    from pt_amd import gltf_export
    p = str(tmp_path / "cube.gltf")
    gltf_export.save_gltf(cube, p)
    dump = str(tmp_path / "frame.f32")
    w, h, spp = 80, 60, 4
    eye, center = (1.5, 2.0, 3.0), (0.0, 0.0, 0.0)
    _run(["--scene", p, "--width", str(w), "--height", str(h), "--spp", str(spp), "--frames", "2", "--dump-result", dump,
          "--eye", *map(str, eye), "--center", *map(str, center)])
    got = np.fromfile(dump, dtype=np.float32).reshape(h, w, 3)
    want, _ = _oracle_frame(cube, w, h, spp, frame=1, eye=eye, center=center)
    assert np.array_equal(got, want)
    assert want.std() > 0.01     # the textured cube is actually in view


def test_gltf_export_import_roundtrip(tmp_path, cube, cornell):
    """CPU: what gltf_export writes, the Python importer reads back to identical arrays and tables."""
    from pt_amd import gltf, gltf_export
    from pt_amd.scene import Scene
    for name, scene in (("cube", cube), ("cornell", cornell)):
        p = str(tmp_path / f"{name}.gltf")
        gltf_export.save_gltf(scene, p)
        s = Scene()
        gltf.load_from_file(s, p)
        a, b = s.models[0], scene.models[0]
        assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.indices, b.indices)
        assert [(m.index_count, m.first_index, m.first_vertex) for m in a.meshes] == [(m.index_count, m.first_index, m.first_vertex) for m in b.meshes]
        for ia, ib in zip(s.instances(), scene.instances()):
            assert np.array_equal(ia.world, ib.world)
        ma, mb = s.tables.material_array(), scene.tables.material_array()
        for f in ("albedoFactor", "emissiveFactor", "useAlbedoMap", "albedoMapIndex"):
            assert np.array_equal(ma[f], mb[f])
        assert all(np.array_equal(x, y) for x, y in zip(s.tables.images, scene.tables.images))


@pytest.mark.gpu
def test_cpp_render_over_several_contexts_equals_one(tmp_path, cornell):
    """pt_render --gpus 3 --rehearse: three renderers (contexts, streams, replicated scene and BVH) on the one
    GPU of the test box, frames dealt round-robin, images summed by pt_accum_allreduce.  Every frame is the
    frame a single renderer traces (same seeds), so the summed image equals the single renderer's up to the
    order of the float additions, holds all seven frames in its alpha, and the ray counts agree exactly."""
    p = str(tmp_path / "cornell.ptscene")
    cornell.save_ptscene(p)
    w, h, spp, frames = 64, 48, 4, 7
    common = ["--scene", p, "--width", str(w), "--height", str(h), "--spp", str(spp), "--frames", str(frames)]
    one, three = str(tmp_path / "one.f32"), str(tmp_path / "three.f32")
    s1 = _run(common + ["--dump-accum", one])
    s3 = _run(common + ["--gpus", "3", "--rehearse", "--dump-accum", three, "--output", str(tmp_path / "img.ppm")])
    a = np.fromfile(one, dtype=np.float32).reshape(h * w, 4)
    b = np.fromfile(three, dtype=np.float32).reshape(h * w, 4)
    assert s3["gpus"] == 3 and s1["rays"] == s3["rays"] and s1["paths"] == s3["paths"] == frames * w * h * spp
    assert np.all(a[:, 3] == frames) and np.all(b[:, 3] == frames)
    assert np.allclose(a[:, :3], b[:, :3], rtol=1e-6, atol=1e-7)
    assert a[:, :3].max() > 0.1


@pytest.mark.gpu
def test_accum_allreduce_is_deterministic_and_complete(cornell):
    """pt_accum_allreduce through the C ABI: four contexts with different images; afterwards all four hold
    the same bits, the sum in context order (checked against numpy adding in that order), also for a
    pixel count that does not divide by the number of contexts, and a second call sums again."""
    import torch
    from pt_amd import hip
    n, n_pixels = 4, 1000 * 7 + 3
    # distinct devices where the box has them (real peer access over xGMI); on a one-GPU box the contexts share device 0
    n_dev = torch.cuda.device_count()
    ctxs = [hip.Context(k % n_dev) for k in range(n)]
    try:
        rng = np.random.default_rng(5)
        imgs = [rng.uniform(0, 4, size=(n_pixels, 4)).astype(np.float32) for _ in range(n)]
        ptrs = [c.upload(im) for c, im in zip(ctxs, imgs)]
        hip.accum_allreduce(ctxs, ptrs, n_pixels)
        want = imgs[0].copy()
        for im in imgs[1:]:
            want += im
        got = [c.download(p, n_pixels * 16, np.float32).reshape(n_pixels, 4) for c, p in zip(ctxs, ptrs)]
        for g in got:
            assert np.array_equal(g, want)
        hip.accum_allreduce(ctxs, ptrs, n_pixels)
        want2 = want.copy()
        for _ in range(n - 1):
            want2 += want
        for c, p in zip(ctxs, ptrs):
            assert np.array_equal(c.download(p, n_pixels * 16, np.float32).reshape(n_pixels, 4), want2)
        same_dev = [k for k in range(n) if k % n_dev == 0][:2]
        if len(same_dev) == 2:
            two = [ctxs[k] for k in same_dev]
            with pytest.raises(hip.PtError):
                hip.accum_allreduce(two, [ptrs[same_dev[0]], ptrs[same_dev[0]]], n_pixels)
            with pytest.raises(hip.PtError):          # images that overlap without being the same pointer
                hip.accum_allreduce(two, [ptrs[same_dev[0]], ptrs[same_dev[0]] + 64], n_pixels)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the command starts its two ranks itself (here both on the box's one
    GPU, all-reduce over gloo: PT_BENCH_REHEARSAL=1) and prints one line that describes what ran: two ranks, every
    frame of both in the reduced image, twice the paths of one rank."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PT_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scene", "cornell", "--width", "320",
           "--height", "180", "--spp", "16", "--frames-in-flight", "1", "--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    par = j["config"]["parallelism"]
    assert j["n_gpus"] == 2 and par["world_size"] == 2 and par["backend"] == "gloo" and j["scaling"] == "weak"
    assert j["frames_accumulated"] == 4 and par["steps_total"] == 4
    assert [r["rank"] for r in par["ranks"]] == [0, 1] and all(r["rehearsal_shared_device"] for r in par["ranks"])
    assert abs(j["paths_per_s"] * j["time_to_image_s"] - 4 * 320 * 180 * 16) < 1.0
    # asking for more ranks than the launcher provides is refused, not silently reduced
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    q = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=120)
    assert q.returncode != 0 and "WORLD_SIZE=1" in q.stderr
