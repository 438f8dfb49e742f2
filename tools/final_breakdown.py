#!/usr/bin/env python3
"""Where the RTNW final scene's frame time goes: renders the scene with one ingredient removed at a
time (marble noise, smoke volumes, specular materials, the spheres) and prints kernel times.
Development tool; images of the edited scenes are not compared with anything."""
import json
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests._desc_tools import OwnedDesc


def timed(g, nx, ny, spp, **kw):
    sc = pt.Scene(g)
    buf = pt.DeviceBuffer(nx * ny * 12)
    sc.render_device(buf.ptr, nx, ny, 1, 25, **kw)
    pt.synchronize()
    best = 1e9
    for _ in range(2):
        sc.render_device(buf.ptr, nx, ny, spp, 25, **kw)
        pt.synchronize()
        best = min(best, sc.stats().kernel_ms)
    return best


def main():
    nx, ny, spp = 1920, 1080, 16
    accel = abi.ACCEL_LBVH if "--list" not in sys.argv else abi.ACCEL_LIST
    pt.set_device(0)

    def base():
        g = OwnedDesc(pt.builtin_scene("final", 16 / 9))
        g.scalars["accel"] = accel
        return g

    def const_tex(g):
        return next(i for i, t in enumerate(g.lists["textures"]) if t.type == abi.TEX_CONSTANT)

    out = {}
    g = base(); g.finalize(); out["as is"] = timed(g, nx, ny, spp)
    g = base()
    for t in g.lists["textures"]:
        if t.type == abi.TEX_NOISE:
            t.type = abi.TEX_CONSTANT; t.color[0] = t.color[1] = t.color[2] = 0.5
    g.finalize(); out["noise -> constant"] = timed(g, nx, ny, spp)
    g = base(); g.lists["world"] = [o for o in g.lists["world"] if o.type != abi.OBJ_MEDIUM]; g.lists["media"] = []
    g.finalize(); out["no media"] = timed(g, nx, ny, spp)
    g = base()
    for m in g.lists["media"]:
        m.density = 1e-30                     # the search still evaluates both volumes, but no ray ever scatters in them
    g.finalize(); out["media that never scatter"] = timed(g, nx, ny, spp)
    g = base()
    g.lists["media"][1].density = 1e-30
    g.finalize(); out["fog never scatters, blue smoke as is"] = timed(g, nx, ny, spp)
    g = base()
    for m in g.lists["materials"]:
        if m.type in (abi.MAT_METAL, abi.MAT_DIELECTRIC):
            m.type = abi.MAT_LAMBERTIAN; m.texture = const_tex(g)
    g.finalize(); out["metal, glass -> lambertian"] = timed(g, nx, ny, spp)
    g = base()
    g.lists["world"] = [o for o in g.lists["world"] if o.type != abi.OBJ_MEDIUM]; g.lists["media"] = []
    for t in g.lists["textures"]:
        if t.type == abi.TEX_NOISE:
            t.type = abi.TEX_CONSTANT
    for m in g.lists["materials"]:
        if m.type in (abi.MAT_METAL, abi.MAT_DIELECTRIC, abi.MAT_ISOTROPIC):
            m.type = abi.MAT_LAMBERTIAN; m.texture = const_tex(g)
    g.finalize(); out["no media, no noise, all lambertian"] = timed(g, nx, ny, spp)
    g.lists["world"] = g.lists["world"][:2]
    g.finalize(); out["boxes and lamp only, lambertian"] = timed(g, nx, ny, spp)
    for k, v in out.items():
        print("%-40s %8.2f ms  %7.0f Mpaths/s" % (k, v, nx * ny * spp / v / 1e3))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
