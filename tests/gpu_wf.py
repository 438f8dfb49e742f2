"""Wavefront scheduler bring-up (not collected by pytest)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests import _oracle
orc = _oracle.load()
for name, aspect, nx, ny, ns in [("cornell", 1.0, 96, 96, 8), ("random", 2.0, 120, 60, 4), ("simple_spheres", 2.0, 80, 40, 8),
                                 ("cornell_spheres", 1.0, 64, 64, 8), ("simple_light", 2.0, 80, 40, 8), ("final", 16 / 9, 96, 54, 8)]:
    d = pt.builtin_scene(name, aspect)
    sc = pt.Scene(d)
    img = sc.render(nx, ny, ns, pipeline=abi.PIPELINE_WAVEFRONT); st = sc.stats()
    ref = orc.render(d, nx, ny, ns, threads=16)
    print("%-16s wavefront bit-exact %s  launches %d  %.2f ms" % (name, np.array_equal(img.view(np.uint32), ref.view(np.uint32)), st.n_launches, st.kernel_ms), flush=True)
d = pt.builtin_scene("cornell", 1.0); sc = pt.Scene(d)
for ns in (16, 64, 256):
    sc.render(1024, 1024, ns, pipeline=abi.PIPELINE_WAVEFRONT); st = sc.stats()
    sc.render(1024, 1024, ns); sf = sc.stats()
    print("cornell 1024^2 x %d: wavefront %.2f ms (%d launches) %.1f Mpaths/s | fused %.2f ms %.1f Mpaths/s" %
          (ns, st.kernel_ms, st.n_launches, 1024 * 1024 * ns / st.kernel_ms / 1e3, sf.kernel_ms, 1024 * 1024 * ns / sf.kernel_ms / 1e3), flush=True)
a = sc.render(256, 256, 32, pipeline=abi.PIPELINE_WAVEFRONT, shard_index=1, shard_count=3); b = sc.render(256, 256, 32, shard_index=1, shard_count=3)
print("sharded wavefront == fused:", np.array_equal(a, b))
a = sc.render(37, 21, 3, 0, pipeline=abi.PIPELINE_WAVEFRONT); b = sc.render(37, 21, 3, 0)
print("depth 0 wavefront == fused:", np.array_equal(a, b))
