"""Host-side Python mirror of the path tracer's hot-path API: ctypes binding of include/pt_hip.h, scene and table model, glTF import."""
from . import abi, camera, scene  # noqa: F401
