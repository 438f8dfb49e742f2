#!/usr/bin/env python3
"""Where the sphere scene's frame time goes: renders random_scene through the library's tree with one
ingredient replaced at a time (specular materials, the checker, moving spheres) and prints kernel times.
Development tool; images of the edited scenes are not compared with anything."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests._desc_tools import OwnedDesc
def timed(g, nx, ny, spp, **kw):
    sc = pt.Scene(g); buf = pt.DeviceBuffer(nx*ny*12)
    sc.render_device(buf.ptr, nx, ny, 1, 25, **kw); pt.synchronize()
    best = 1e9
    for _ in range(2):
        sc.render_device(buf.ptr, nx, ny, spp, 25, **kw); pt.synchronize(); best = min(best, sc.stats().kernel_ms)
    st = sc.stats()
    return best, st.schedule
pt.set_device(0)
nx, ny, spp = 1920, 1080, 64
def base():
    g = OwnedDesc(pt.builtin_scene("random", 16/9)); g.scalars["accel"] = abi.ACCEL_LBVH; return g
def const_tex(g): return next(i for i,t in enumerate(g.lists["textures"]) if t.type == abi.TEX_CONSTANT)
out = {}
g = base(); g.finalize(); out["as is"] = timed(g, nx, ny, spp)
g = base()
for m in g.lists["materials"]:
    if m.type in (abi.MAT_METAL, abi.MAT_DIELECTRIC): m.type = abi.MAT_LAMBERTIAN; m.texture = const_tex(g)
g.finalize(); out["metal, glass -> lambertian"] = timed(g, nx, ny, spp)
g = base()
for t in g.lists["textures"]:
    if t.type == abi.TEX_CHECKER: t.type = abi.TEX_CONSTANT; t.color[0]=t.color[1]=t.color[2]=0.5
g.finalize(); out["checker -> constant"] = timed(g, nx, ny, spp)
g = base()
for m in g.lists["materials"]:
    if m.type in (abi.MAT_METAL, abi.MAT_DIELECTRIC): m.type = abi.MAT_LAMBERTIAN; m.texture = const_tex(g)
for t in g.lists["textures"]:
    if t.type == abi.TEX_CHECKER: t.type = abi.TEX_CONSTANT; t.color[0]=t.color[1]=t.color[2]=0.5
g.finalize(); out["all lambertian constant"] = timed(g, nx, ny, spp)
g = base()
for s in g.lists["shapes"]:
    if s.type == abi.SHAPE_MOVING_SPHERE:
        s.type = abi.SHAPE_SPHERE; s.p[3] = s.p[8]
g.finalize(); out["moving -> static spheres"] = timed(g, nx, ny, spp)
for k,(v,sch) in out.items(): print("%-32s %7.2f ms %6.0f Mpaths/s (schedule %d)" % (k, v, nx*ny*spp/v/1e3, sch))
