#!/usr/bin/env python3
"""First run of the staged pipeline in chunks: small frames against the fused kernel, then the headline frame timed.  Development tool."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

pt.set_device(0)
for name in ("cornell", "random"):
    d = pt.builtin_scene(name, 1.0)
    sc = pt.Scene(d)
    for nx, ny, ns in ((16, 16, 5), (72, 40, 7), (3, 2, 5)):
        ref = sc.render(nx, ny, ns)
        for chunk in (1, 2, 3, ns - 1, ns):
            img = sc.render(nx, ny, ns, pipeline=abi.PIPELINE_WAVEFRONT, chunk_samples=chunk)
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            print(name, nx, ny, ns, "chunk", chunk, "chunks", sc.stats().n_chunks, "equal" if same else "DIFFERENT", flush=True)
            if not same:
                sys.exit(1)
print("small frames equal", flush=True)
