"""Host-side Python mirror of the path tracer's hot-path API (see DESIGN.md at the repo root)."""
from . import abi, camera, scene  # noqa: F401
