#!/usr/bin/env python3
"""Renders the Cornell frame on the staged (wavefront) pipeline, for profiling:
    python tools/wavefront_render.py [SPP]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = pt.Scene(pt.builtin_scene("cornell", 1.0))
buf = pt.DeviceBuffer(3 * 1024 * 1024 * 4)
for _ in range(2):
    sc.render_device(buf.ptr, 1024, 1024, spp, pipeline=abi.PIPELINE_WAVEFRONT)
    pt.synchronize()
st = sc.stats()
print("staged pipeline, 1024x1024 x%d: %.2f ms, %.0f Mpaths/s, %d launches" % (spp, st.kernel_ms, st.paths / st.kernel_ms / 1e3, st.n_launches))
