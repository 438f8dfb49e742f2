"""Not collected by pytest: one 1/8 shard of the headline frame, for profiling the low-occupancy case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpupathtracer_amd as pt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sched = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = pt.builtin_scene("cornell", 1.0); sc = pt.Scene(d)
buf = pt.DeviceBuffer(1024 * 1024 * 12)
sc.render_device(buf.ptr, 1024, 1024, spp, 25, shard_index=0, shard_count=shards, schedule=sched); pt.synchronize()
print("1/%d shard, %d spp: %.2f ms" % (shards, spp, sc.stats().kernel_ms))
