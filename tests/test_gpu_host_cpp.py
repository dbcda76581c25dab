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
