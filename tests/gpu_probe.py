"""First GPU bring-up probe (not collected by pytest): parity numbers + timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi
from tests import _oracle

orc = _oracle.load()
print("devices", pt.device_count(), flush=True)
for name, aspect, nx, ny, ns in [("cornell", 1.0, 128, 128, 16), ("random", 2.0, 200, 100, 16),
                                 ("simple_spheres", 2.0, 200, 100, 16), ("cornell_spheres", 1.0, 128, 128, 16),
                                 ("simple_light", 2.0, 200, 100, 16), ("final", 16 / 9, 192, 108, 16)]:
    d = pt.builtin_scene(name, aspect)
    t0 = time.time(); ref = orc.render(d, nx, ny, ns, threads=16); t_cpu = time.time() - t0
    sc = pt.Scene(d)
    t0 = time.time(); img = sc.render(nx, ny, ns); t_gpu = time.time() - t0
    st = sc.stats()
    exact = float(np.mean(img.view(np.uint32) == ref.view(np.uint32)))
    print("%-16s psnr %.2f dB rmse %.3e bit-exact-frac %.4f  cpu %.2fs gpu %.3fs kernel %.3f ms" %
          (name, _oracle.psnr(img, ref), _oracle.rmse(img, ref), exact, t_cpu, t_gpu, st.kernel_ms), flush=True)
    rays = orc.record_rays(d, nx // 2, ny // 2, 6, 100000)
    ids_o, t_o = orc.trace_closest(d, rays)
    ids_g, t_g = sc.trace_closest(rays)
    same = (ids_o == ids_g) & (t_o.view(np.uint32) == t_g.view(np.uint32))
    print("   closest-hit: %d rays, id+t bit-exact %d (%.5f), id mismatches %d" %
          (len(rays), same.sum(), same.mean(), (ids_o != ids_g).sum()), flush=True)
    if name in ("random",):
        d2 = pt.builtin_scene(name, aspect, accel=abi.ACCEL_LBVH)
        sc2 = pt.Scene(d2)
        ids_l, t_l = sc2.trace_closest(rays)
        same = (ids_o == ids_l) & (t_o.view(np.uint32) == t_l.view(np.uint32))
        print("   LBVH closest-hit: id+t bit-exact %.5f, id mismatches %d, nodes %d" %
              (same.mean(), (ids_o != ids_l).sum(), len(sc2.lbvh_nodes())), flush=True)
        img2 = sc2.render(nx, ny, ns)
        print("   LBVH image vs list image identical: %s  kernel %.3f ms" % (np.array_equal(img2, img), sc2.stats().kernel_ms), flush=True)
# throughput probe
d = pt.builtin_scene("cornell", 1.0)
sc = pt.Scene(d)
for ns in (16, 64, 256):
    sc.render(1024, 1024, ns)
    st = sc.stats()
    print("cornell 1024^2 x %d spp: kernel %.2f ms -> %.1f Mpaths/s" % (ns, st.kernel_ms, 1024 * 1024 * ns / st.kernel_ms / 1e3), flush=True)
