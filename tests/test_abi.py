"""The C-ABI library loads, exports every symbol include/pt_hip.h declares, and the Python-side
struct layouts match the header byte for byte (checked by compiling the header with gcc)."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

from pt_amd import abi, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pt_hip.h")


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_last_error_is_a_string_without_a_gpu():
    lib = hip.load_library()
    assert isinstance(lib.pt_last_error(), bytes)


def test_struct_layouts_match_header():
    fields = {
        "PtMaterial": ["albedoFactor", "metallicFactor", "emissiveFactor", "useEmissiveMap", "useAlbedoMap", "albedoMapIndex", "emissiveMapIndex"],
        "PtGeometryNode": ["vertexBufferDeviceAddress", "indexBufferDeviceAddress", "materialIndex"],
        "PtInstance": ["transform", "instanceCustomIndexAndMask", "sbtOffsetAndFlags", "accelerationStructureReference"],
        "PtTextureDesc": ["rgba8_dev", "width", "height"],
        "PtHit": ["instance", "primitive", "u", "v"],
        "PtBvhNode": ["lo0", "hi0", "lo1", "hi1", "c0", "c1"],
        "PtAsInfo": ["node_count", "prim_count", "root_lo", "root_hi", "delta", "build_ms"],
        "PtTraceStats": ["rays_per_depth", "rays", "paths"],
        "PtTraceProfile": ["intersect_launches", "intersect_rays", "intersect_ms", "primary_launches", "primary_rays", "primary_ms"],
    }
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for s, fs in fields.items():
        src.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fs:
            src.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    src.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "l.c")
        open(c, "w").write("\n".join(src))
        subprocess.check_call(["gcc", "-std=c11", "-o", os.path.join(d, "l"), c])
        out = subprocess.check_output([os.path.join(d, "l")]).decode().split("\n")
    got = dict(line.split() for line in out if line)
    py = {"PtMaterial": abi.MATERIAL, "PtGeometryNode": abi.GEOMETRY_NODE, "PtInstance": abi.INSTANCE,
          "PtTextureDesc": abi.TEXTURE_DESC, "PtHit": abi.HIT, "PtBvhNode": abi.BVH_NODE,
          "PtAsInfo": hip.AS_INFO, "PtTraceStats": hip.TRACE_STATS, "PtTraceProfile": hip.TRACE_PROFILE}
    rename = {"instanceCustomIndexAndMask": "customIndexAndMask", "accelerationStructureReference": "blasRef", "rgba8_dev": "rgba8"}
    for s, dt in py.items():
        assert int(got[s]) == dt.itemsize, s
        for f in fields[s]:
            assert int(got[f"{s}.{f}"]) == dt.fields[rename.get(f, f)][1], f"{s}.{f}"
    # the reference's own sizes (shaders/bindless.glsl std140 / gpu_resources.hpp)
    assert abi.MATERIAL.itemsize == 96 and abi.GEOMETRY_NODE.itemsize == 32 and abi.INSTANCE.itemsize == 64 and abi.VERTEX.itemsize == 32


def test_product_package_does_not_touch_the_oracle():
    """The shipped path must never import, load or fall back to the CPU oracle."""
    pkg = os.path.join(ROOT, "path-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "import oracle" not in text and "pt_oracle.h" not in text, os.path.join(dirpath, f)
