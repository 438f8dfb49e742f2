"""Probe: HIP runtime coexistence with torch, shard emulation timings (not collected by pytest)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpupathtracer_amd as pt
d = pt.builtin_scene("cornell", 1.0)
sc = pt.Scene(d)
img0 = sc.render(256, 256, 16)
print("lib alone ok", img0.mean(), flush=True)
# shard emulation: one shard of 8 on this GPU vs the full frame
for ns in (256, 1024):
    sc.render(1024, 1024, ns); full = sc.stats().kernel_ms
    times = []
    for si in range(8):
        sc.render(1024, 1024, ns, shard_index=si, shard_count=8); times.append(sc.stats().kernel_ms)
    print("ns %d full %.2f ms; 1/8 shards %s ms; est. 8-GPU efficiency %.3f" % (ns, full, ["%.2f" % t for t in times], full / (8 * max(times))), flush=True)
    for tile in ((8, 8), (16, 16), (32, 32), (64, 64)):
        sc.render(1024, 1024, ns, shard_index=0, shard_count=8, tile=tile); t8 = sc.stats().kernel_ms
        sc.render(1024, 1024, ns, shard_index=0, shard_count=2, tile=tile); t2 = sc.stats().kernel_ms
        print("   tile %s: shard 1/8 %.2f ms (eff %.3f), shard 1/2 %.2f ms (eff %.3f)" % (tile, t8, full / (8 * t8), t2, full / (2 * t2)), flush=True)
import torch
print("torch", torch.__version__, torch.cuda.is_available(), flush=True)
x = torch.zeros(256 * 256 * 3, device="cuda", dtype=torch.float32)
torch.cuda.synchronize()
img1 = sc.render(256, 256, 16)
print("lib after torch init ok:", np.array_equal(img0, img1), flush=True)
try:
    sc.render_device(x.data_ptr(), 256, 256, 16)
    st = sc.stats()
    torch.cuda.synchronize()
    y = x.cpu().numpy().reshape(256, 256, 3)
    print("render into torch tensor:", np.array_equal(y, img0), "kernel ms", st.kernel_ms, flush=True)
except Exception as e:
    print("render into torch tensor failed:", e, flush=True)
