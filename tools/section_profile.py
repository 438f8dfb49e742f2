#!/usr/bin/env python3
"""Section timers of a kernel built with -DPT_PROFILE (tools/build_exp.sh prof -DPT_PROFILE <unit>; run with
PT_LIBPTCORE_PATH=tools/exp/libptcore_prof.so): where a wave's time goes, and how many of its lanes had work there.
    python tools/section_profile.py random 1920 1080 64 lbvh [schedule N] [chunk N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

NAMES = ["tree node step", "tree leaf step", "rest of the bounce (hit completion, material, scatter, end of path)", "next pixel slot", "chunk record",
         "camera ray + begin walk", "6", "7", "8", "9", "10", "nobody needed the section"]
name, nx, ny, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rest = sys.argv[5:]
accel = abi.ACCEL_LBVH if "lbvh" in rest else abi.ACCEL_LIST
kw = {}
if "schedule" in rest:
    kw["schedule"] = int(rest[rest.index("schedule") + 1])
if "chunk" in rest:
    kw["chunk_samples"] = int(rest[rest.index("chunk") + 1])
pt.set_device(0)
if name.startswith("trimesh"):
    d = pt.trimesh_scene(int(name[7:] or 1000), nx / ny)
    d.c.accel = accel
elif name == "final_earth":
    sys.path.insert(0, os.path.join(ROOT, "bench"))
    from configs import synthetic_earthmap
    d = pt.builtin_scene("final", nx / ny, earth_rgb=synthetic_earthmap(), accel=accel)
else:
    d = pt.builtin_scene(name, nx / ny, accel=accel)
lib = pt.library()
sc = pt.Scene(d)
buf = pt.DeviceBuffer(nx * ny * 12)
sc.render_device(buf.ptr, nx, ny, max(1, spp // 8), 25, **kw)
pt.synchronize()
n = lib.pt_debug_profile(None, 1)
out = (C.c_ulonglong * (3 * n))()
sc.render_device(buf.ptr, nx, ny, spp, 25, **kw)
pt.synchronize()
st = sc.stats()
lib.pt_debug_profile(out, 0)
tot = sum(out[3 * k] for k in range(n)) or 1
print("%s %dx%dx%d %s: %.2f ms, schedule %d, %.0f Mpaths/s (with the timers in)" % (name, nx, ny, spp, " ".join(rest), st.kernel_ms, st.schedule, st.paths / st.kernel_ms / 1e3))
print("%-30s %8s %12s %12s %10s" % ("section", "time %", "runs", "ticks/run", "lanes/run"))
for k in range(n):
    t, l, r = out[3 * k], out[3 * k + 1], out[3 * k + 2]
    if t or r:
        print("%-30s %7.2f%% %12d %12.1f %10.1f" % (NAMES[k], 100.0 * t / tot, r, t / max(r, 1), l / max(r, 1)))
