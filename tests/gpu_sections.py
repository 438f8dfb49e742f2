import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
lib = pt.api.library()
out = (ctypes.c_ulonglong * 16)()
for name, nx, ny, spp in (("cornell", 1024, 1024, 64), ("random", 960, 540, 4), ("final", 960, 540, 8)):
    desc = pt.builtin_scene(name, nx / ny)
    sc = pt.Scene(desc)
    buf = pt.DeviceBuffer(nx * ny * 12)
    sc.render_device(buf.ptr, nx, ny, spp, 25); pt.synchronize()
    lib.pt_debug_sections(out, 1)
    sc.render_device(buf.ptr, nx, ny, spp, 25); pt.synchronize()
    lib.pt_debug_sections(out, 0)
    v = list(out)
    tot = v[0] + v[2] + v[4]
    print(name, "kernel ms %.2f" % sc.stats().kernel_ms)
    for k, nm in enumerate(("fresh", "world_hit", "shade")):
        print("  %-10s cycles %5.1f%%  lanes/entry %.1f  (lane-entries %d)" % (nm, 100.0 * v[2*k] / tot, 0, v[2*k+1]))
    print("  loop iterations (wave) %d, lanes per iteration %.1f" % (v[6], v[7] / max(1, v[6])))
