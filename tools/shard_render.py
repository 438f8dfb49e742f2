#!/usr/bin/env python3
"""Renders one tile shard of the Cornell frame on one GPU (what one of N ranks does), for profiling:
    python tools/shard_render.py SHARD_COUNT [SPP]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpupathtracer_amd as pt

count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sc = pt.Scene(pt.builtin_scene("cornell", 1.0))
buf = pt.DeviceBuffer(3 * 1024 * 1024 * 4)
for _ in range(2):
    sc.render_device(buf.ptr, 1024, 1024, spp, shard_index=0, shard_count=count, tile=(8, 8))
    pt.synchronize()
print("1/%d shard, %d spp: %.2f ms" % (count, spp, sc.stats().kernel_ms))
