#!/usr/bin/env python3
"""Scene creation time of the 2 M-triangle field (upload, Morton codes, sort, tree, 4-wide form), three times.  Development tool."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
pt.set_device(0)
d = pt.trimesh_scene(1000, 16 / 9)
d.c.accel = abi.ACCEL_LBVH
for k in range(3):
    t0 = time.perf_counter()
    sc = pt.Scene(d)
    pt.synchronize()
    print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    sc.close()
