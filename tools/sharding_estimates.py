#!/usr/bin/env python3
"""What each way of dividing the headline frame over N GPUs can reach, estimated on one GPU: the kernel time of one
rank's share (the slowest tile shard of batch 0, which is the reference's own stream) against the whole frame's.
Estimated N-GPU efficiency = T(whole frame) / (N x T(one rank's share)); the exchange (one RCCL reduce of a 12.6 MB
frame, well under a millisecond over xGMI) is not in it.  Writes one JSON object.
    python tools/sharding_estimates.py [spp] > profiles/r03/sharding_estimates.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpupathtracer_amd as pt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nx = ny = 1024
pt.set_device(0)
sc = pt.Scene(pt.builtin_scene("cornell", 1.0))
buf = pt.DeviceBuffer(nx * ny * 12)


def kernel_ms(**kw):
    best = None
    for _ in range(3):
        sc.render_device(buf.ptr, nx, ny, kw.pop("ns", spp) if "ns" in kw else spp, 25, **kw)
        pt.synchronize()
        t = sc.stats().kernel_ms
        best = t if best is None else min(best, t)
    return best


def share_ms(tiles, batches):
    """the slowest of the `tiles` tile shards, tracing spp / batches samples"""
    worst = 0.0
    for t in range(tiles):
        kw = dict(shard_index=t, shard_count=tiles)
        if batches > 1:
            kw.update(batch_index=0, batch_count=batches, raw_sum=1)
        ms = None
        for _ in range(2):
            sc.render_device(buf.ptr, nx, ny, spp // batches, 25, **kw)
            pt.synchronize()
            k = sc.stats().kernel_ms
            ms = k if ms is None else min(ms, k)
        worst = max(worst, ms)
    return worst


whole = kernel_ms()
out = {"workload": "cornell_box %dx%d %dspp depth25" % (nx, ny, spp), "whole_frame_kernel_ms": whole, "sharding": []}
for n in (2, 4, 8):
    for tiles in sorted({n, 1, 2, 4} & {d for d in (1, 2, 4, 8) if n % d == 0}, reverse=True):
        batches = n // tiles
        ms = share_ms(tiles, batches)
        out["sharding"].append({"n_gpus": n, "tile_shards": tiles, "sample_batches": batches, "slowest_share_kernel_ms": ms,
                                "estimated_efficiency": whole / (n * ms),
                                "image": "the reference's, bit for bit" if batches == 1 else "within the stated RMSE tolerance of sampling noise (batch 0 is the reference's stream)"})
print(json.dumps(out, indent=1))
