"""Counting build of the oracle (oracle/liboracle_count.so, -DPT_ORACLE_COUNT): closest-hit
searches, RNG draws, diffuse scatters and arithmetic per path.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

import gpupathtracer_amd as pt
from gpupathtracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["flop", "libm", "iop", "draws", "paths", "world_hits", "rect_tests", "sphere_tests", "triangle_tests",
          "aabb_tests", "diffuse_scatters", "accepted_hits", "tree_nodes", "tree_leaves"]
_LIB = None


def _load():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "oracle", "liboracle_count.so")
        src = os.path.join(ROOT, "oracle", "pt_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_count.so"])
        _LIB = C.CDLL(so)
        _LIB.oracle_render.argtypes = [C.POINTER(A.PtSceneDesc), C.POINTER(A.PtRenderParams), C.c_void_p, C.c_int]
        _LIB.oracle_work_counters.argtypes = [C.c_void_p, C.c_int]
        _LIB.oracle_count_through_tree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_count_tree_form.argtypes = [C.c_int]
    return _LIB


def counters(desc, nx, ny, ns, max_depth=25, threads=8, tree=None, wide=False):
    """tree: None (the work of the reference's own flat walk), or (nodes, root, cull, prefix, trust): the
    library's exported LBVH as an (n, 16) uint32 array -- searches are then priced as a replay of that
    tree over world[:prefix] plus the in-order walk of the rest; trust = let the replay's hit drive the
    path (for worlds too long for the flat walk).  wide: price the search as the 4-wide walk of that tree (every
    second level folded away; tree_nodes then counts 64-byte wide nodes)."""
    lib = _load()
    if tree is not None:
        nodes, root, cull, prefix, trust = tree
        nodes = np.ascontiguousarray(nodes, np.uint32)
        lib.oracle_count_through_tree(nodes.ctypes.data, int(root), int(cull), int(prefix), int(trust))
        lib.oracle_count_tree_form(1 if wide else 0)
    else:
        lib.oracle_count_through_tree(None, 0, 0, 0, 0)
    p = A.PtRenderParams(nx=nx, ny=ny, ns=ns, max_depth=max_depth, shard_index=0, shard_count=1, tile_w=8, tile_h=8)
    img = np.zeros((ny, nx, 3), np.float32)
    lib.oracle_render(desc.ptr, C.byref(p), img.ctypes.data, threads)
    cnt = (C.c_uint64 * len(FIELDS))()
    lib.oracle_work_counters(cnt, len(FIELDS))
    lib.oracle_count_through_tree(None, 0, 0, 0, 0)
    lib.oracle_count_tree_form(0)
    return dict(zip(FIELDS, [int(v) for v in cnt])), img


def run(name, aspect, nx, ny, ns):
    d = pt.builtin_scene(name, aspect)
    c, _ = counters(d, nx, ny, ns)
    n = float(nx * ny * ns)
    assert c["paths"] == nx * ny * ns
    return {"world_hits_per_path": c["world_hits"] / n, "rng_draws_per_path": c["draws"] / n,
            "diffuse_scatters_per_path": c["diffuse_scatters"] / n, "flop_per_path": c["flop"] / n,
            "rect_tests_per_path": c["rect_tests"] / n}
