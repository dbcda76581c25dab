"""Camera uniform of the reference: CameraUniformData { viewInverse, projInverse }
(include/renderer.hpp:31-35), filled by Renderer::InitializeCamera (source/renderer.cpp:172-193):
perspectiveRH_ZO(radians(60), w/h, 0.1, 1000) with proj[1][1] negated, lookAt((0,1,3),(0,1,0),(0,1,0)).
Returned as 32 float32: viewInverse then projInverse, both column-major like glm::mat4.

The projection entries are formed in binary32 as glm does and inverted in closed form in binary64;
the C++ host (path-tracer_amd/host/renderer.cpp) evaluates the same expressions, so both hosts hand
the kernels identical bits.
"""
import math

import numpy as np

f32 = np.float32


def camera_uniform(width, height, eye=(0.0, 1.0, 3.0), center=(0.0, 1.0, 0.0), fov_deg=60.0):
    aspect = f32(width) / f32(height)
    t = f32(math.tan(float(f32(fov_deg)) * 0.017453292519943295 / 2.0))
    z_near, z_far = f32(0.1), f32(1000.0)
    p00 = f32(1.0) / (aspect * t)
    p11 = -(f32(1.0) / t)
    p22 = z_far / (z_near - z_far)
    p32 = -(z_far * z_near) / (z_far - z_near)
    pi = np.zeros(16, dtype=f32)          # pi[c * 4 + r]
    pi[0 * 4 + 0] = f32(1.0 / float(p00))
    pi[1 * 4 + 1] = f32(1.0 / float(p11))
    pi[3 * 4 + 2] = f32(-1.0)
    pi[2 * 4 + 3] = f32(1.0 / float(p32))
    pi[3 * 4 + 3] = f32(float(p22) / float(p32))
    e = [float(f32(v)) for v in eye]
    c = [float(f32(v)) for v in center]
    f = [c[k] - e[k] for k in range(3)]
    ln = math.sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2])
    f = [v / ln for v in f]
    up = (0.0, 1.0, 0.0)
    s = [f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]]
    ln = math.sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2])
    s = [v / ln for v in s]
    u = [s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]]
    vi = np.zeros(16, dtype=f32)
    for r in range(3):
        vi[0 * 4 + r] = f32(s[r])
        vi[1 * 4 + r] = f32(u[r])
        vi[2 * 4 + r] = f32(-f[r])
        vi[3 * 4 + r] = f32(e[r])
    vi[15] = f32(1.0)
    return np.concatenate([vi, pi]).astype(f32)
