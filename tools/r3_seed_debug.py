#!/usr/bin/env python3
"""One textured fuzz world whose frame differs from the oracle's under a tree: which pixels, which samples, which searches.
    python tools/r3_seed_debug.py 100401"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests import _oracle
import test_gpu_parity as T

seed = int(sys.argv[1])
oracle = _oracle.load()
pt.set_device(0)
rng = np.random.default_rng(5000 + seed)
tex = rng.integers(0, 256, (16, 32, 3), dtype=np.uint8)
g = T._random_world(rng, pt.builtin_scene("final", 16 / 9, earth_rgb=tex), textured=True)
cam = pt.builtin_scene("simple_spheres", 2.0).c.camera
g.camera = [float(cam[k]) * (12.0 if k < 3 else 1.0) for k in range(24)]
g.camera[23] = 0.0 if seed % 2 == 0 else float(rng.uniform(0.01, 2.0))
g.scalars["ambient_type"] = 2
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
nx, ny = 64, 32
for accel in (abi.ACCEL_LIST, abi.ACCEL_LBVH_BINARY, abi.ACCEL_LBVH_WIDE):
    g.scalars["accel"] = accel
    g.finalize()
    sc = pt.Scene(g)
    print("== accel", accel, "world entries", g.c.n_world, "shapes", g.c.n_shapes, "media", g.c.n_media, "bvhs", g.c.n_bvhs, "lbvh form", sc.lbvh_form(), flush=True)
    for ns in (1, 2, 3, 4, 5, 6):
        ref = oracle.render(g, nx, ny, ns)
        for sched in (abi.SCHEDULE_TILES, abi.SCHEDULE_REFILL):
            img = sc.render(nx, ny, ns, schedule=sched)
            bad = np.argwhere((bits(img) != bits(ref)).any(axis=2))
            if len(bad):
                print(" ns", ns, "schedule", sched, "pixels that differ:", [tuple(int(v) for v in b) for b in bad][:10], flush=True)
    ref = oracle.render(g, nx, ny, 6)
    img = sc.render(nx, ny, 6)
    bad = np.argwhere((bits(img) != bits(ref)).any(axis=2))
    for (row, col) in bad[:3]:
        pix = int(row) * nx + int(col)
        rays = oracle.record_rays_ns(g, nx, ny, 6, 25, 4096, only_pixel=pix)
        ids_o, t_o = oracle.trace_closest(g, rays)
        ids_g, t_g = sc.trace_closest(rays)
        diff = np.nonzero((ids_o != ids_g) | (bits(t_o) != bits(t_g)))[0]
        print(" pixel", (int(row), int(col)), "rays of its six samples:", len(rays), "searches that differ:", diff.tolist()[:10], flush=True)
        for k in diff[:4]:
            r = rays[k]
            print("   ray", k, "o", r["o"].tolist(), "d", r["d"].tolist(), "time", float(r["time"]), "oracle id/t", int(ids_o[k]), float(t_o[k]), "gpu id/t", int(ids_g[k]), float(t_g[k]), flush=True)
            for ids_ in (int(ids_o[k]), int(ids_g[k])):
                if 0 <= ids_ < g.c.n_shapes:
                    s = g.c.shapes[ids_]
                    print("     shape", ids_, "type", s.type, "material", s.material, "xforms", s.xform_count, "p", [float(v) for v in s.p], flush=True)
        if not len(diff):
            print("   every recorded search agrees: the difference is in the shading (uv?) -- values:", img[row, col].tolist(), ref[row, col].tolist(), flush=True)
