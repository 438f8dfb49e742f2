"""ctypes binding of the C ABI in include/ptcore.h.

Mirrors the seam of the reference's main() (src/ptMain.cu:559-653): build a scene
(``builtin_scene`` ~ the hard-coded scene call at :559), hand it to the device
(``Scene`` ~ serialise + allocate_world_kernel, :560-610) and render (``Scene.render`` ~
render_kernel + cudaMemcpy, :612-649).  Errors surface as ``PtError`` carrying the library's
message, where the reference prints to std::cerr and exits.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_LIB = None


def library_path():
    """The in-tree build; PT_LIBPTCORE_PATH names another build of the same library (kernel experiments)."""
    return os.environ.get("PT_LIBPTCORE_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptcore.so")


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ptcore error %d: %s" % (code, message))
        self.code = code


def library():
    """Loads libptcore.so (once).  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `make -C gpupathtracer_amd/csrc` "
            "(or __graft_entry__.build()); this package has no CPU path" % path)
    lib = C.CDLL(path)
    P = C.POINTER
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_last_error.argtypes = []
    lib.pt_scene_create.argtypes = [P(A.PtSceneDesc), P(C.c_void_p)]
    lib.pt_scene_destroy.argtypes = [C.c_void_p]
    lib.pt_scene_destroy.restype = None
    lib.pt_render.argtypes = [C.c_void_p, P(A.PtRenderParams), C.c_void_p]
    lib.pt_render_device.argtypes = [C.c_void_p, P(A.PtRenderParams), C.c_void_p, C.c_void_p]
    lib.pt_render_device_noalloc.argtypes = [C.c_void_p, P(A.PtRenderParams), C.c_void_p, C.c_void_p]
    lib.pt_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.pt_render_stats.argtypes = [C.c_void_p, P(A.PtRenderStats)]
    if hasattr(lib, "pt_render_stage_stats"):        # (an older build named by PT_LIBPTCORE_PATH for a timing comparison lacks the newer entry points)
        lib.pt_render_stage_stats.argtypes = [C.c_void_p, P(A.PtStageStats)]
    lib.pt_scene_lbvh_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.pt_scene_lbvh_form.argtypes = [C.c_void_p]
    lib.pt_scene_lbvh_wide_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.pt_scene_desc_builtin.argtypes = [C.c_char_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                          P(P(A.PtSceneDesc))]
    lib.pt_scene_desc_trimesh.argtypes = [C.c_int32, C.c_float, P(P(A.PtSceneDesc))]
    lib.pt_scene_desc_free.argtypes = [P(A.PtSceneDesc)]
    if hasattr(lib, "pt_image_decode_jpeg"):
        lib.pt_image_decode_jpeg.argtypes = [C.c_char_p, C.c_size_t, P(C.c_int32), P(C.c_int32), C.c_void_p, C.c_size_t]
    lib.pt_scene_desc_free.restype = None
    lib.pt_render_on_devices.argtypes = [P(A.PtSceneDesc), P(A.PtRenderParams), P(C.c_int32), C.c_int32, C.c_void_p]
    lib.pt_frame_finish_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]
    lib.pt_accum_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.pt_accum_bytes.restype = C.c_size_t
    lib.pt_render_pass.argtypes = [C.c_void_p, P(A.PtRenderParams), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.pt_render_progressive.argtypes = [C.c_void_p, P(A.PtRenderParams), C.c_void_p, C.c_int32, PROGRESS_FN, C.c_void_p]
    lib.pt_memcpy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.pt_median_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.pt_median_filter_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.pt_scene_desc_from_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, P(P(A.PtSceneDesc))]
    lib.pt_scene_desc_to_stream.argtypes = [P(A.PtSceneDesc), C.c_void_p, C.c_size_t, P(C.c_size_t), P(C.c_size_t)]
    lib.pt_device_count.argtypes = []
    lib.pt_set_device.argtypes = [C.c_int]
    lib.pt_device_synchronize.argtypes = []
    lib.pt_device_malloc.argtypes = [C.c_size_t, P(C.c_void_p)]
    lib.pt_device_free.argtypes = [C.c_void_p]
    lib.pt_device_free.restype = None
    lib.pt_memcpy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.pt_comm_unique_id.argtypes = [C.c_void_p]
    lib.pt_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]
    lib.pt_comm_reduce_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.pt_comm_barrier.argtypes = [C.c_void_p, C.c_void_p]
    lib.pt_comm_allreduce_max_f64.argtypes = [C.c_void_p, P(C.c_double), C.c_void_p]
    lib.pt_comm_destroy.argtypes = [C.c_void_p]
    lib.pt_comm_destroy.restype = None
    lib.pt_libm_eval.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PtError(rc, library().pt_last_error().decode("utf-8", "replace"))


def device_count():
    return int(library().pt_device_count())


def set_device(index):
    _check(library().pt_set_device(int(index)))


def decode_jpeg(data):
    """A baseline JPEG file (bytes) -> (ny, nx, 3) uint8 texels, as the reference's stbi_load decodes it."""
    lib = library()
    blob = bytes(data)
    nx, ny = C.c_int32(0), C.c_int32(0)
    _check(lib.pt_image_decode_jpeg(blob, len(blob), C.byref(nx), C.byref(ny), None, 0))
    out = np.empty((ny.value, nx.value, 3), np.uint8)
    _check(lib.pt_image_decode_jpeg(blob, len(blob), C.byref(nx), C.byref(ny), out.ctypes.data, out.nbytes))
    return out


def synchronize():
    _check(library().pt_device_synchronize())


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.c_void_p)


def render_on_devices(desc, device_ids, nx, ny, ns, max_depth=25, **kw):
    """The whole frame traced by several GPUs of this process (tile shards added on the host)."""
    p = make_params(nx, ny, ns, max_depth, **kw)
    ids = (C.c_int32 * len(device_ids))(*device_ids)
    out = np.zeros((ny, nx, 3), np.float32)
    _check(library().pt_render_on_devices(desc.ptr, C.byref(p), ids, len(device_ids), out.ctypes.data))
    return out


def frame_finish(d_frame_ptr, n_floats, total_samples, stream=None):
    _check(library().pt_frame_finish_device(C.c_void_p(d_frame_ptr), n_floats, total_samples, C.c_void_p(stream)))


def accum_bytes(nx, ny):
    return int(library().pt_accum_bytes(nx, ny))


class DeviceBuffer:
    """Device memory owned by the library's HIP runtime."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(library().pt_device_malloc(nbytes, C.byref(p)))
        self.ptr = p.value
        self.nbytes = nbytes

    def from_host(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes == self.nbytes
        _check(library().pt_memcpy_to_device(C.c_void_p(self.ptr), a.ctypes.data, a.nbytes))

    def to_host(self, dtype=np.float32):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        _check(library().pt_memcpy_to_host(out.ctypes.data, C.c_void_p(self.ptr), self.nbytes))
        return out

    def close(self):
        if self.ptr:
            library().pt_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of one rank (one process per GPU)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(library().pt_comm_unique_id(buf))
        return bytes(buf.raw)

    def __init__(self, unique_id, rank, world):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(library().pt_comm_init(buf, rank, world, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    def reduce_frame(self, dev_ptr, n_floats, root=0, stream=None):
        _check(library().pt_comm_reduce_frame(self._h, C.c_void_p(dev_ptr), n_floats, root,
                                              C.c_void_p(stream) if stream else None))

    def barrier(self, stream=None):
        _check(library().pt_comm_barrier(self._h, C.c_void_p(stream) if stream else None))

    def allreduce_max(self, value, stream=None):
        v = C.c_double(float(value))
        _check(library().pt_comm_allreduce_max_f64(self._h, C.byref(v), C.c_void_p(stream) if stream else None))
        return float(v.value)

    def close(self):
        if self._h is not None:
            library().pt_comm_destroy(self._h)
            self._h = None


def libm_eval(fn, x, on_device):
    """fn: 0 sin, 1 cos, 2 log, 3 x**5, 4/5 sine/cosine of the combined evaluation, 6 asin, 7 atan2
    (x then holds (y, x) pairs, shape (n, 2)) -- the library's libm replacements on float32 inputs;
    8/9 the kernels' in-range square root and sqrtf, 10/11 their in-range quotient and n / d on (n, d)
    pairs (on the host all four are the plain operation); 12/13 sine/cosine of the combined evaluation in its
    form for arguments in [0, 120)."""
    x = np.ascontiguousarray(x, np.float32)
    pairs = int(fn) in (7, 10, 11)
    n = x.size // 2 if pairs else x.size
    out = np.empty(n, np.float32)
    _check(library().pt_libm_eval(int(fn), x.ctypes.data, n, out.ctypes.data, 1 if on_device else 0))
    return out.reshape(x.shape[:-1]) if pairs else out.reshape(x.shape)


class SceneDesc:
    """Owns a library-allocated PtSceneDesc (one of the reference's scenes)."""

    def __init__(self, ptr):
        self._ptr = ptr

    @property
    def ptr(self):
        return self._ptr

    @property
    def c(self):
        return self._ptr.contents

    def close(self):
        if self._ptr is not None:
            library().pt_scene_desc_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def builtin_scene(name, aspect, earth_rgb=None, accel=None):
    """One of simple_spheres, simple_light, random, cornell, cornell_spheres, final, trimesh
    (src/ptMain.cu:192-421).  earth_rgb: optional HxWx3 uint8 texels for ``final``."""
    lib = library()
    out = C.POINTER(A.PtSceneDesc)()
    if earth_rgb is not None:
        tex = np.ascontiguousarray(earth_rgb, dtype=np.uint8)
        assert tex.ndim == 3 and tex.shape[2] == 3
        _check(lib.pt_scene_desc_builtin(name.encode(), float(aspect), tex.ctypes.data, tex.shape[1], tex.shape[0],
                                         C.byref(out)))
    else:
        _check(lib.pt_scene_desc_builtin(name.encode(), float(aspect), None, 0, 0, C.byref(out)))
    if accel is not None:
        out.contents.accel = int(accel)
    return SceneDesc(out)


def median_filter(image, size=3):
    """medianFilter3x3 / medianFilter2x2 of the reference (src/ptMain.cu:726-788) on the GPU."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    assert img.ndim == 3 and img.shape[2] == 3
    out = np.empty_like(img)
    _check(library().pt_median_filter(img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0], int(size)))
    return out


def scene_from_stream(data, flags=0, accel=None):
    """A scene serialised by the reference (world, light shapes, camera, ambient light;
    src/ptMain.cu:560-572) as a description."""
    lib = library()
    blob = bytes(data)
    out = C.POINTER(A.PtSceneDesc)()
    _check(lib.pt_scene_desc_from_stream(blob, len(blob), int(flags), C.byref(out)))
    if accel is not None:
        out.contents.accel = int(accel)
    return SceneDesc(out)


def scene_to_stream(desc):
    """(bytes, [world, lights, camera, ambient] section sizes) in the reference's stream format."""
    lib = library()
    size = C.c_size_t(0)
    sections = (C.c_size_t * 4)()
    _check(lib.pt_scene_desc_to_stream(desc.ptr, None, 0, C.byref(size), sections))
    buf = C.create_string_buffer(size.value)
    _check(lib.pt_scene_desc_to_stream(desc.ptr, buf, size.value, C.byref(size), sections))
    return buf.raw[:size.value], [int(v) for v in sections]


def trimesh_scene(grid_n, aspect):
    lib = library()
    out = C.POINTER(A.PtSceneDesc)()
    _check(lib.pt_scene_desc_trimesh(int(grid_n), float(aspect), C.byref(out)))
    return SceneDesc(out)


class RenderStats:
    def __init__(self, s):
        self.kernel_ms = float(s.kernel_ms)
        self.total_ms = float(s.total_ms)
        self.n_launches = int(s.n_launches)
        self.pipeline = int(s.pipeline)
        self.paths = int(s.paths)
        self.schedule = int(s.schedule)
        self.signature_walk = int(s.signature_walk)
        self.kernel_family = int(s.kernel_family)
        self.n_chunks = int(s.n_chunks)


def make_params(nx, ny, ns, max_depth=25, shard_index=0, shard_count=1, tile=(8, 8), pipeline=0, schedule=0,
                batch_index=0, batch_count=0, raw_sum=0, chunk_samples=0, stage_stats=0, slot_block_tiles=0):
    return A.PtRenderParams(nx=nx, ny=ny, ns=ns, max_depth=max_depth, shard_index=shard_index,
                            shard_count=shard_count, tile_w=tile[0], tile_h=tile[1], pipeline=pipeline, schedule=schedule,
                            batch_index=batch_index, batch_count=batch_count, raw_sum=raw_sum, chunk_samples=chunk_samples,
                            stage_stats=stage_stats, slot_block_tiles=slot_block_tiles)


class Scene:
    """A scene resident on the current HIP device."""

    def __init__(self, desc):
        lib = library()
        h = C.c_void_p()
        ptr = desc.ptr if hasattr(desc, "ptr") else desc
        _check(lib.pt_scene_create(ptr, C.byref(h)))
        self._h = h

    def close(self):
        if self._h is not None:
            library().pt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, nx, ny, ns, max_depth=25, **kw):
        """Returns the (ny, nx, 3) float32 frame on the host; row 0 is the top of the image."""
        p = make_params(nx, ny, ns, max_depth, **kw)
        out = np.empty((ny, nx, 3), np.float32)
        _check(library().pt_render(self._h, C.byref(p), out.ctypes.data))
        return out

    def render_device(self, d_out_ptr, nx, ny, ns, max_depth=25, stream=None, **kw):
        """Queues a render into device memory (d_out_ptr: integer device address of nx*ny*3 floats)."""
        p = make_params(nx, ny, ns, max_depth, **kw)
        _check(library().pt_render_device(self._h, C.byref(p), C.c_void_p(d_out_ptr),
                                          C.c_void_p(stream) if stream else None))

    def render_device_noalloc(self, d_out_ptr, nx, ny, ns, max_depth=25, stream=None, **kw):
        """render_device through the entry point that allocates no device memory (static slot counter)."""
        p = make_params(nx, ny, ns, max_depth, **kw)
        _check(library().pt_render_device_noalloc(self._h, C.byref(p), C.c_void_p(d_out_ptr),
                                                  C.c_void_p(stream) if stream else None))

    def render_pass(self, d_accum_ptr, first_sample, n_samples, d_out_ptr, nx, ny, ns, max_depth=25, stream=None, **kw):
        """Samples [first_sample, first_sample + n_samples) of a render of ns samples (device buffers)."""
        p = make_params(nx, ny, ns, max_depth, **kw)
        _check(library().pt_render_pass(self._h, C.byref(p), C.c_void_p(d_accum_ptr), first_sample, n_samples,
                                        C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def render_progressive(self, nx, ny, ns, pass_samples, progress=None, max_depth=25, **kw):
        """Whole render in passes; progress(done, total) returning True stops it.  Returns (image, cancelled)."""
        p = make_params(nx, ny, ns, max_depth, **kw)
        out = np.zeros((ny, nx, 3), np.float32)
        cb = PROGRESS_FN((lambda done, total, user: 1 if progress(done, total) else 0) if progress else (lambda d, t, u: 0))
        rc = library().pt_render_progressive(self._h, C.byref(p), out.ctypes.data, pass_samples, cb, None)
        if rc == A.ERR_CANCELLED:
            return out, True
        _check(rc)
        return out, False

    def stats(self):
        s = A.PtRenderStats()
        _check(library().pt_render_stats(self._h, C.byref(s)))
        return RenderStats(s)

    def stage_stats(self):
        """Counters of the staged pipeline's most recent render (a dict; see PtStageStats)."""
        s = A.PtStageStats()
        _check(library().pt_render_stage_stats(self._h, C.byref(s)))
        d = {k: int(getattr(s, k)) for k in ("groups", "rounds", "extend_lanes", "extend_waves", "extend_ticks", "shade_ticks", "timed_groups")}
        d["shade_lanes"] = [int(x) for x in s.shade_lanes]
        d["shade_waves"] = [int(x) for x in s.shade_waves]
        d["extend_occupancy"] = d["extend_lanes"] / max(1, 64 * d["extend_waves"])
        d["shade_occupancy"] = sum(d["shade_lanes"]) / max(1, 64 * sum(d["shade_waves"]))
        d["extend_time_share"] = d["extend_ticks"] / max(1, d["extend_ticks"] + d["shade_ticks"])
        return d

    def trace_closest(self, rays):
        """rays: structured/packed array of PtRay (n x 32 bytes).  Returns (prim_id int32[n], t float32[n])."""
        rays = np.ascontiguousarray(rays)
        assert rays.nbytes % 32 == 0
        n = rays.nbytes // 32
        ids = np.empty(n, np.int32)
        ts = np.empty(n, np.float32)
        _check(library().pt_trace_closest(self._h, rays.ctypes.data, n, ids.ctypes.data, ts.ctypes.data))
        return ids, ts

    def lbvh_form(self):
        """0: the searches walk no library-built tree; 2: its binary nodes; 4: its 4-wide form."""
        n = library().pt_scene_lbvh_form(self._h)
        if n < 0:
            _check(n)
        return n

    def lbvh_wide_nodes(self):
        """The 4-wide form of the tree as an (n, 16) uint32 view of its 64-byte nodes (empty if the scene walks none)."""
        n = library().pt_scene_lbvh_wide_nodes(self._h, None, 0)
        if n < 0:
            _check(n)
        buf = np.zeros((max(n, 0), 16), np.uint32)
        if n > 0:
            rc = library().pt_scene_lbvh_wide_nodes(self._h, buf.ctypes.data, buf.nbytes)
            if rc < 0:
                _check(rc)
        return buf

    def lbvh_nodes(self):
        """The library-built LBVH as an (n, 16) uint32 view of its 64-byte nodes (empty if none)."""
        n = library().pt_scene_lbvh_nodes(self._h, None, 0)
        if n < 0:
            _check(n)
        buf = np.zeros((max(n, 0), 16), np.uint32)
        if n > 0:
            rc = library().pt_scene_lbvh_nodes(self._h, buf.ctypes.data, buf.nbytes)
            if rc < 0:
                _check(rc)
        return buf


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("time", np.float32), ("seed", np.uint32)])
