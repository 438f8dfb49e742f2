"""MI355X-native radiance core for rweyrauch/gpupathtracer's hot path.

The product is the C-ABI shared library ``libptcore.so`` (declared in ``include/ptcore.h``,
built from ``gpupathtracer_amd/csrc`` for gfx950).  This package is a thin ctypes binding over
that ABI for tests and benchmarks; it contains no rendering code of its own and raises if the
library has not been built -- there is no CPU fallback.
"""
from .api import (  # noqa: F401
    PtError, Scene, SceneDesc, RenderStats, DeviceBuffer, Comm, builtin_scene, trimesh_scene,
    scene_from_stream, scene_to_stream, median_filter, accum_bytes, frame_finish, render_on_devices,
    device_count, set_device, synchronize, libm_eval, library, library_path,
)
from . import _abi as abi  # noqa: F401

__all__ = ["PtError", "Scene", "SceneDesc", "RenderStats", "DeviceBuffer", "Comm", "builtin_scene",
           "trimesh_scene", "scene_from_stream", "scene_to_stream", "median_filter", "accum_bytes", "frame_finish", "render_on_devices", "device_count", "set_device", "synchronize", "libm_eval", "library",
           "library_path", "abi"]
