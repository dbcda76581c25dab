"""Camera uniform of the reference: CameraUniformData { viewInverse, projInverse }
(include/renderer.hpp:31-35), filled by Renderer::InitializeCamera (source/renderer.cpp:172-193):
perspectiveRH_ZO(radians(60), w/h, 0.1, 1000) with proj[1][1] negated, lookAt((0,1,3),(0,1,0),(0,1,0)).
Returned as 32 float32: viewInverse then projInverse, both column-major like glm::mat4.
"""
import numpy as np

f32 = np.float32


def perspective_rh_zo(fovy, aspect, z_near, z_far):
    t = f32(np.tan(f32(fovy) / f32(2)))
    m = np.zeros((4, 4), dtype=f32)  # math convention m[row, col]
    m[0, 0] = f32(1) / (f32(aspect) * t)
    m[1, 1] = f32(1) / t
    m[2, 2] = f32(z_far) / (f32(z_near) - f32(z_far))
    m[3, 2] = -f32(1)
    m[2, 3] = -(f32(z_far) * f32(z_near)) / (f32(z_far) - f32(z_near))
    return m


def look_at(eye, center, up):
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    fwd = center - eye
    fwd /= np.linalg.norm(fwd)
    s = np.cross(fwd, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, fwd)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -fwd
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, fwd @ eye
    return m


def camera_uniform(width, height, eye=(0.0, 1.0, 3.0), center=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=60.0):
    aspect = f32(width) / f32(height)
    proj = perspective_rh_zo(np.radians(f32(fov_deg)), aspect, 0.1, 1000.0)
    proj[1, 1] *= f32(-1)
    proj_inv = np.linalg.inv(proj.astype(np.float64)).astype(f32)
    view_inv = np.linalg.inv(look_at(eye, center, up)).astype(f32)
    out = np.empty(32, dtype=f32)
    out[:16] = view_inv.T.reshape(16)   # column-major
    out[16:] = proj_inv.T.reshape(16)
    return out
