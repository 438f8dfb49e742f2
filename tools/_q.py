import sys; sys.path.insert(0,'.')
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
a = pt.Scene(pt.builtin_scene("final", 16/9)).render(1920, 1080, 16)
b = pt.Scene(pt.builtin_scene("final", 16/9, accel=abi.ACCEL_LBVH)).render(1920, 1080, 16)
diff = (a.view(np.uint32) != b.view(np.uint32)).reshape(-1, 3).any(axis=1)
print("pixels differing:", int(diff.sum()), "of", diff.size, "paths", diff.size*16)
mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64))**2))
print("mse", mse, "psnr", 10*np.log10(1.0/mse) if mse > 0 else "inf")
sc0 = pt.Scene(pt.builtin_scene("final", 16/9)); sc1 = pt.Scene(pt.builtin_scene("final", 16/9, accel=abi.ACCEL_LBVH))
buf = pt.DeviceBuffer(3*1920*1080*4)
for name, s in (("list", sc0), ("lbvh-prefix", sc1)):
    for sched in (1, 2):
        for rep in range(2):
            s.render_device(buf.ptr, 1920,1080,16, schedule=sched); pt.synchronize(); t = s.stats().kernel_ms
        print(name, "sched", sched, "%.2f ms" % t, flush=True)
a = sc0.render(1920, 1080, 64); b = sc1.render(1920, 1080, 64)
print("64 spp: pixels differing:", int((a.view(np.uint32) != b.view(np.uint32)).reshape(-1,3).any(axis=1).sum()))
