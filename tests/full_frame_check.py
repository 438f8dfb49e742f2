#!/usr/bin/env python3
"""The headline frame against the oracle at full size: Cornell 1024x1024 at 4096 spp, every channel of
every pixel compared bit for bit.  Two minutes of host time on sixteen threads, so it is a script and
not part of the collected tests:

    python tests/full_frame_check.py [--scene cornell] [--size 1024] [--ny N] [--spp 4096] [--lbvh] [--out result.json]

The oracle renders the frame in tile shards (progress line after each); the GPU renders it in one launch."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt                     # noqa: E402
from tests import _oracle                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--size", type=int, default=1024, help="frame width (and height unless --ny is given)")
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--trimesh", type=int, default=0, help="the height field of 2*N*N triangles in the Cornell room instead of --scene")
    ap.add_argument("--wavefront", action="store_true", help="the staged scheduler instead of the fused one")
    ap.add_argument("--lbvh", action="store_true", help="the GPU walks the library's tree; the oracle always walks the list")
    ap.add_argument("--earthmap", action="store_true", help="the final scene with the synthetic 512x256 earth map on its tenth object")
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--shards", type=int, default=32)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    oracle = _oracle.load()
    pt.set_device(0)
    nx, ny = a.size, (a.ny or a.size)
    if a.trimesh:
        d = pt.trimesh_scene(a.trimesh, nx / ny)
    elif a.earthmap:
        sys.path.insert(0, os.path.join(ROOT, "bench"))
        import configs as bench_configs
        d = pt.builtin_scene(a.scene, nx / ny, earth_rgb=bench_configs.synthetic_earthmap())
    else:
        d = pt.builtin_scene(a.scene, nx / ny)
    if a.lbvh:
        d.c.accel = pt.abi.ACCEL_LBVH
    sc = pt.Scene(d)
    t0 = time.perf_counter()
    kw = {"pipeline": pt.abi.PIPELINE_WAVEFRONT} if a.wavefront else {}
    gpu = sc.render(nx, ny, a.spp, **kw)
    gpu_s = time.perf_counter() - t0
    print("gpu frame in %.2f s (kernel %.1f ms)" % (gpu_s, sc.stats().kernel_ms), flush=True)
    cpu = np.zeros_like(gpu)
    t0 = time.perf_counter()
    for k in range(a.shards):
        cpu += oracle.render(d, nx, ny, a.spp, shard_index=k, shard_count=a.shards, threads=a.threads)
        print("oracle shard %d/%d, %.0f s" % (k + 1, a.shards, time.perf_counter() - t0), flush=True)
    cpu_s = time.perf_counter() - t0
    same = gpu.view(np.uint32) == cpu.view(np.uint32)
    res = {"scene": ("trimesh %d" % a.trimesh) if a.trimesh else a.scene + (" + synthetic earth map" if a.earthmap else ""), "nx": nx, "ny": ny, "spp": a.spp, "gpu_accel": "lbvh" if a.lbvh else "list", "gpu_pipeline": "wavefront" if a.wavefront else "fused", "gpu_schedule": sc.stats().schedule, "values": int(same.size), "values_different": int((~same).sum()),
           "oracle_seconds": cpu_s, "oracle_threads": a.threads, "oracle_Mpaths_per_s": nx * ny * a.spp / cpu_s / 1e6,
           "gpu_kernel_ms": sc.stats().kernel_ms, "frame_mean": float(gpu.mean())}
    print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    return 0 if same.all() else 1


if __name__ == "__main__":
    sys.exit(main())
