#!/usr/bin/env python3
"""Kernel time of one scene at one size: python tools/quick_time.py cornell 1024 1024 256 [lbvh] [schedule N] [chunk SAMPLES] [shards N] [bands N] [wavefront]
Development tool."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

name, nx, ny, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rest = sys.argv[5:]
accel = abi.ACCEL_LBVH if "lbvh" in rest else abi.ACCEL_LBVH_BINARY if "binary" in rest else abi.ACCEL_LBVH_WIDE if "wide" in rest else abi.ACCEL_LIST
kw = {}
if "schedule" in rest:
    kw["schedule"] = int(rest[rest.index("schedule") + 1])
if "chunk" in rest:                       # with `schedule 2|3`: the persistent-lane schedules in chunks of that many samples (>= spp: never)
    kw.setdefault("schedule", abi.SCHEDULE_CHUNKS)
    kw["chunk_samples"] = int(rest[rest.index("chunk") + 1])
if "wavefront" in rest:
    kw["pipeline"] = abi.PIPELINE_WAVEFRONT
if "bands" in rest:                     # pixel slots of the shared walk: blocks of N wave tiles over eight counters (< 0: one counter)
    kw["slot_block_tiles"] = int(rest[rest.index("bands") + 1])
if "shards" in rest:
    kw["shard_count"] = int(rest[rest.index("shards") + 1])
pt.set_device(0)
if name.startswith("trimesh"):
    d = pt.trimesh_scene(int(name[7:] or 1000), nx / ny)
    d.c.accel = accel
else:
    d = pt.builtin_scene(name, nx / ny, accel=accel)
sc = pt.Scene(d)
buf = pt.DeviceBuffer(nx * ny * 12)
sc.render_device(buf.ptr, nx, ny, max(1, spp // 8), 25, **kw)
pt.synchronize()
ms = []
for _ in range(3):
    sc.render_device(buf.ptr, nx, ny, spp, 25, **kw)
    pt.synchronize()
    ms.append(sc.stats().kernel_ms)
st = sc.stats()
print("%s %dx%dx%d %s: %.3f ms  %.0f Mpaths/s  (schedule %d, chunks %d, signature %d, family %d, all %s)" %
      (name, nx, ny, spp, " ".join(rest), min(ms), st.paths / min(ms) / 1e3, st.schedule, st.n_chunks, st.signature_walk, st.kernel_family, ["%.2f" % m for m in ms]), flush=True)
