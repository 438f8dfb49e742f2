#!/usr/bin/env python3
"""Algorithmic work per path of the benchmark configurations -> bench/work_model.json.

Renders a bounded sample of each configuration with the counting build of the CPU checker
(oracle/liboracle_count.so, -DPT_ORACLE_COUNT) and divides its counters by the paths traced.
The counters follow the reference's own evaluation (flat world list, every accepted hit record
completed), which is the work the roofline figure of bench.py prices the GPU kernel against.
Convention: one FLOP = one fp32 add/sub/mul/compare/min/max/div/sqrt, no FMA credit; libm calls
and integer operations are reported beside it, not inside it.

Runs on the CPU; needs no GPU (scene descriptors come from the library's host-side factory).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # the repository root
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt                     # noqa: E402
from gpupathtracer_amd import _abi as A            # noqa: E402

FIELDS = ["flop", "libm", "iop", "draws", "paths", "world_hits", "rect_tests", "sphere_tests", "triangle_tests",
          "aabb_tests", "diffuse_scatters", "accepted_hits"]
# (key, scene, aspect, nx, ny, spp of the sample, note)
CONFIGS = [
    ("cornell", "cornell", 1.0, 256, 256, 64, "BASELINE configs[2]: Cornell box, depth 25"),
    ("random", "random", 16 / 9, 320, 180, 8, "BASELINE configs[1]: RTIOW, the reference's flat list of 487 spheres"),
    ("final", "final", 16 / 9, 320, 180, 8, "RTNW final scene, reference BVH over 400 boxes + media"),
    ("simple_light", "simple_light", 2.0, 200, 100, 16, "two noise spheres, sphere + rectangle lights"),
]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_count.so"])
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_count.so"))
    lib.oracle_render.argtypes = [C.POINTER(A.PtSceneDesc), C.POINTER(A.PtRenderParams), C.c_void_p, C.c_int]
    lib.oracle_work_counters.argtypes = [C.c_void_p, C.c_int]
    out = {"convention": "FLOP = fp32 add/sub/mul/compare/min/max/div/sqrt as the reference evaluates them "
                         "(flat lists, hit record completed on every accepted hit); no FMA credit; libm and integer ops separate",
           "generator": "tests/make_work_model.py (oracle/pt_oracle.c built with -DPT_ORACLE_COUNT)", "configs": {}}
    for key, scene, aspect, nx, ny, ns, note in CONFIGS:
        desc = pt.builtin_scene(scene, aspect)
        p = A.PtRenderParams(nx=nx, ny=ny, ns=ns, max_depth=25, shard_index=0, shard_count=1, tile_w=8, tile_h=8)
        img = np.zeros((ny, nx, 3), np.float32)
        lib.oracle_render(desc.ptr, C.byref(p), img.ctypes.data, 8)
        cnt = (C.c_uint64 * len(FIELDS))()
        lib.oracle_work_counters(cnt, len(FIELDS))
        c = dict(zip(FIELDS, [int(v) for v in cnt]))
        paths = c["paths"]
        assert paths == nx * ny * ns
        out["configs"][key] = {
            "note": note, "sample": "%dx%d at %d spp, depth 25" % (nx, ny, ns),
            "flop_per_path": c["flop"] / paths, "libm_calls_per_path": c["libm"] / paths,
            "integer_ops_per_path": c["iop"] / paths, "rng_draws_per_path": c["draws"] / paths,
            "closest_hit_searches_per_path": c["world_hits"] / paths, "rect_tests_per_path": c["rect_tests"] / paths,
            "sphere_tests_per_path": c["sphere_tests"] / paths, "triangle_tests_per_path": c["triangle_tests"] / paths,
            "aabb_tests_per_path": c["aabb_tests"] / paths, "diffuse_scatters_per_path": c["diffuse_scatters"] / paths,
            "accepted_hits_per_path": c["accepted_hits"] / paths,
        }
        print(key, json.dumps(out["configs"][key]), flush=True)
    with open(os.path.join(ROOT, "bench", "work_model.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
