"""Quick GPU check used while optimising (not a test): bit-exactness on all scenes + Cornell throughput."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests import _oracle
orc = _oracle.load()
ok = True
for name, aspect, nx, ny, ns in [("cornell", 1.0, 96, 96, 8), ("random", 2.0, 120, 60, 4), ("simple_spheres", 2.0, 80, 40, 8),
                                 ("cornell_spheres", 1.0, 64, 64, 8), ("simple_light", 2.0, 80, 40, 8), ("final", 16 / 9, 96, 54, 8)]:
    d = pt.builtin_scene(name, aspect)
    sc = pt.Scene(d)
    ref = orc.render(d, nx, ny, ns, threads=16)
    e = True
    for sched in (1, 2):
        img = sc.render(nx, ny, ns, schedule=sched)
        e &= np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    ok &= e
    print("%-16s bit-exact %s" % (name, e), flush=True)
d = pt.builtin_scene("cornell", 1.0); sc = pt.Scene(d)
for ns in (64, 1024):
  for sched in (1, 2):
    sc.render(1024, 1024, ns, schedule=sched); st = sc.stats()
    print("cornell 1024^2 x %d sched %d: %.2f ms  %.1f Mpaths/s" % (ns, sched, st.kernel_ms, 1024 * 1024 * ns / st.kernel_ms / 1e3), flush=True)
for sched in (1, 2):
    sc.render(1024, 1024, 1024, shard_index=0, shard_count=8, schedule=sched)
    print("1/8 shard sched %d: %.2f ms" % (sched, sc.stats().kernel_ms), flush=True)
sc.render(1024, 1024, 1024, shard_index=0, shard_count=8); t8 = sc.stats().kernel_ms
sc.render(1024, 1024, 1024); t1 = sc.stats().kernel_ms
print("1/8 shard %.2f ms -> est. 8-GPU efficiency %.3f" % (t8, t1 / (8 * t8)), flush=True)
d = pt.builtin_scene("random", 16 / 9, accel=abi.ACCEL_LBVH); sc = pt.Scene(d)
sc.render(1920, 1080, 16); st = sc.stats()
print("rtiow lbvh 1920x1080 x16: %.2f ms  %.1f Mpaths/s" % (st.kernel_ms, 1920 * 1080 * 16 / st.kernel_ms / 1e3), flush=True)
d = pt.builtin_scene("random", 16 / 9); sc = pt.Scene(d)
sc.render(1920, 1080, 4); st = sc.stats()
print("rtiow list 1920x1080 x4: %.2f ms  %.1f Mpaths/s" % (st.kernel_ms, 1920 * 1080 * 4 / st.kernel_ms / 1e3), flush=True)
d = pt.builtin_scene("final", 16 / 9); sc = pt.Scene(d)
sc.render(1920, 1080, 16); st = sc.stats()
print("final 1920x1080 x16: %.2f ms  %.1f Mpaths/s" % (st.kernel_ms, 1920 * 1080 * 16 / st.kernel_ms / 1e3), flush=True)
print("ALL BIT-EXACT" if ok else "MISMATCH", flush=True)
