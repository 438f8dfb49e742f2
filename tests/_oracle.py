"""ctypes binding of the CPU checker (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from gpupathtracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        P = C.POINTER
        lib.oracle_render.argtypes = [P(A.PtSceneDesc), P(A.PtRenderParams), C.c_void_p, C.c_int]
        lib.oracle_trace_closest.argtypes = [P(A.PtSceneDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.oracle_trace_records.argtypes = [P(A.PtSceneDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.oracle_record_rays.argtypes = [P(A.PtSceneDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.oracle_record_rays.restype = C.c_size_t
        lib.oracle_record_rays_ns.argtypes = [P(A.PtSceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.oracle_record_rays_ns.restype = C.c_size_t
        lib.oracle_camera_rays.argtypes = [P(A.PtSceneDesc), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.oracle_pcg_stream.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        lib.oracle_simple_rng_stream.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_void_p]
        lib.oracle_noise.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.oracle_texture_values.argtypes = [P(A.PtSceneDesc), C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.oracle_aabb_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        lib.oracle_libm.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.oracle_median_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.oracle_fnv1a.argtypes = [C.c_void_p, C.c_size_t]
        lib.oracle_fnv1a.restype = C.c_uint64

    def render(self, desc, nx, ny, ns, max_depth=25, shard_index=0, shard_count=1, tile=(8, 8), threads=8):
        p = A.PtRenderParams(nx=nx, ny=ny, ns=ns, max_depth=max_depth, shard_index=shard_index,
                             shard_count=shard_count, tile_w=tile[0], tile_h=tile[1])
        out = np.zeros((ny, nx, 3), np.float32)
        self.lib.oracle_render(desc.ptr, C.byref(p), out.ctypes.data, threads)
        return out

    def trace_closest(self, desc, rays):
        rays = np.ascontiguousarray(rays)
        n = rays.nbytes // 32
        ids = np.empty(n, np.int32)
        ts = np.empty(n, np.float32)
        self.lib.oracle_trace_closest(desc.ptr, rays.ctypes.data, n, ids.ctypes.data, ts.ctypes.data)
        return ids, ts

    def trace_records(self, desc, rays):
        rays = np.ascontiguousarray(rays)
        n = rays.nbytes // 32
        ids = np.empty(n, np.int32)
        rec = np.empty((n, 9), np.float32)
        self.lib.oracle_trace_records(desc.ptr, rays.ctypes.data, n, ids.ctypes.data, rec.ctypes.data)
        return ids, rec

    def record_rays(self, desc, nx, ny, max_depth, capacity):
        from gpupathtracer_amd.api import RAY_DTYPE
        rays = np.zeros(capacity, RAY_DTYPE)
        n = self.lib.oracle_record_rays(desc.ptr, nx, ny, max_depth, rays.ctypes.data, capacity)
        return rays[:n].copy()

    def record_rays_ns(self, desc, nx, ny, ns, max_depth, capacity, only_pixel=-1):
        """Rays of the first ns samples (the pixel's random stream running on), of one pixel if only_pixel >= 0."""
        from gpupathtracer_amd.api import RAY_DTYPE
        rays = np.zeros(capacity, RAY_DTYPE)
        n = self.lib.oracle_record_rays_ns(desc.ptr, nx, ny, ns, max_depth, only_pixel, rays.ctypes.data, capacity)
        return rays[:n].copy()

    def camera_rays(self, desc, nx, ny, pixels):
        """First camera ray of each pixel index in `pixels` (no world search)."""
        from gpupathtracer_amd.api import RAY_DTYPE
        px = np.ascontiguousarray(pixels, np.uint32)
        rays = np.zeros(px.size, RAY_DTYPE)
        self.lib.oracle_camera_rays(desc.ptr, nx, ny, px.ctypes.data, px.size, rays.ctypes.data)
        return rays

    def median_filter(self, image, size=3):
        img = np.ascontiguousarray(image, np.float32)
        out = np.empty_like(img)
        self.lib.oracle_median_filter(img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0], size)
        return out

    def pcg_stream(self, seed, n):
        out = np.empty(n, np.float32)
        self.lib.oracle_pcg_stream(seed, n, out.ctypes.data)
        return out

    def simple_rng_stream(self, s0, s1, n):
        out = np.empty(n, np.float32)
        self.lib.oracle_simple_rng_stream(s0, s1, n, out.ctypes.data)
        return out

    def noise(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        n = pts.shape[0]
        a = np.empty(n, np.float32)
        b = np.empty(n, np.float32)
        self.lib.oracle_noise(pts.ctypes.data, n, a.ctypes.data, b.ctypes.data)
        return a, b

    def texture_values(self, desc, tex, uvp):
        uvp = np.ascontiguousarray(uvp, np.float32)
        n = uvp.shape[0]
        out = np.empty((n, 3), np.float32)
        self.lib.oracle_texture_values(desc.ptr, tex, uvp.ctypes.data, n, out.ctypes.data)
        return out

    def aabb_hits(self, boxes, rays, tmin, tmax):
        boxes = np.ascontiguousarray(boxes, np.float32)
        rays = np.ascontiguousarray(rays)
        n = boxes.shape[0]
        out = np.empty(n, np.int32)
        self.lib.oracle_aabb_hits(boxes.ctypes.data, rays.ctypes.data, n, tmin, tmax, out.ctypes.data)
        return out

    def libm(self, fn, x):
        """fn 0 sin, 1 cos, 2 log, 3 x**5, 6 asin, 7 atan2 of (y, x) pairs (shape (n, 2))."""
        x = np.ascontiguousarray(x, np.float32)
        n = x.size // 2 if fn == 7 else x.size
        out = np.empty(n, np.float32)
        self.lib.oracle_libm(fn, x.ctypes.data, n, out.ctypes.data)
        return out.reshape(x.shape[:-1]) if fn == 7 else out.reshape(x.shape)

    def fnv1a(self, arr):
        arr = np.ascontiguousarray(arr)
        return int(self.lib.oracle_fnv1a(arr.ctypes.data, arr.nbytes))


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "pt_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        _LIB = Oracle(C.CDLL(path))
    return _LIB


def psnr(a, b, peak=1.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    if mse == 0:
        return float("inf")
    return 10.0 * np.log10(peak * peak / mse)


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=(0, 1))).max())
