#!/usr/bin/env python3
"""Writes tests/golden/cfg1_cornell_512_16spp.npz: outputs of the CPU oracle for BASELINE config 1
(Cornell box, 512x512, 16 samples, frame 0, seed = TEA(pixel, sample index), depth 10, reference
camera): the bounce-0 (instance, primitive) image, a 128x128 crop of the radiance image and the
SHA-256 of the full float32 radiance image.  The reference itself ships no golden data for this path,
so these pin the build against itself (oracle run here vs oracle / HIP run elsewhere)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from pt_amd import camera  # noqa: E402
from pt_amd.scene import Scene  # noqa: E402

W = H = 512
SPP = 16
scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
orc = O.OracleScene(scene)
cam = camera.camera_uniform(W, H)
res, st = orc.render_frame(cam, W, H, frame=0, spp=SPP, threads=os.cpu_count())
o = np.zeros((H * W, 3), np.float32)
d = np.zeros((H * W, 3), np.float32)
for y in range(H):
    for x in range(W):
        o[y * W + x], d[y * W + x] = O.primary_ray(cam, W, H, x, y)
hits, t = orc.tlas.intersect(o, d)
ids = np.stack([hits["instance"].astype(np.int32), hits["primitive"].astype(np.int32)], axis=1).reshape(H, W, 2)
ids[ids > 255] = 255   # miss marker
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cfg1_cornell_512_16spp.npz"),
                    bounce0_ids=ids.astype(np.uint8), crop=res[192:320, 192:320].copy(),
                    sha256=np.frombuffer(hashlib.sha256(res.tobytes()).digest(), dtype=np.uint8),
                    rays=np.uint64(st["rays"]), mean=res.mean(axis=(0, 1)).astype(np.float64))
print("rays", int(st["rays"]), "mean", res.mean(axis=(0, 1)), hashlib.sha256(res.tobytes()).hexdigest())
