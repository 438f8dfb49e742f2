#!/usr/bin/env python3
"""The other benchmark configurations on one GPU (the headline, Cornell 1024^2 x 4096, is bench.py):
RTIOW 1920x1080 with the LBVH, the RTNW final scene, and the 2M-triangle height field in the
Cornell enclosure.  Prints one JSON object per configuration; --quick cuts samples for a smoke run."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt                     # noqa: E402
from gpupathtracer_amd import abi                  # noqa: E402


def synthetic_earthmap():
    """512x256 RGB8, the size of the reference's earthmap.jpg (the JPEG itself does not travel and
    is not decoded here): a fixed arithmetic pattern, the same as `ptrace --earthmap synthetic`."""
    import numpy as np
    j, i = np.mgrid[0:256, 0:512]
    return np.stack([(i * 7 + j * 13) & 255, (i ^ (j * 3)) & 255, ((i * j) >> 3) & 255], -1).astype(np.uint8)


PEAK_VALU_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12     # fp32 vector, one add/mul per lane per cycle, no FMA credit (as bench.py)
PEAK_HBM_GBS = 8000.0


def roofline(model_key, bound, paths_per_s):
    """achieved = algorithmic work per path (bench/work_model.json, counted by the instrumented CPU checker; for
    the *_lbvh entries through a replay of the very tree the GPU walks) x measured paths per second."""
    path = os.path.join(ROOT, "bench", "work_model.json")
    if model_key is None or not os.path.exists(path):
        return None
    m = json.load(open(path))["configs"].get(model_key)
    if m is None:
        return None
    if bound == "hbm":
        if "bytes_per_path" not in m:
            return None
        a = m["bytes_per_path"] * paths_per_s / 1e9
        return {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS,
                "bytes_per_path": m["bytes_per_path"], "model": model_key}
    a = m["flop_per_path"] * paths_per_s / 1e12
    return {"bound": "valu", "achieved": a, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_VALU_TFLOPS,
            "flop_per_path": m["flop_per_path"], "model": model_key}


def cpu_baseline(desc, nx, ny, spp, what):
    """The CPU checker (oracle/, a restatement of the reference's host path) on a bounded sample of the same workload,
    on every host thread this job may use.  A reported baseline, not a target."""
    sys.path.insert(0, ROOT)
    from tests import _oracle
    orc = _oracle.load()
    visible = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(visible, share) if share > 0 else visible)
    t0 = time.perf_counter()
    orc.render(desc, nx, ny, spp, 25, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": nx * ny * spp / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port", "seconds": dt,
            "sample": "%s, %dx%d at %d spp, depth 25, OpenMP rows over %d threads" % (what, nx, ny, spp, threads)}


def measured_hbm(pmc_name, kernel_ms_now):
    """Bytes a launch really moved, from the committed rocprofv3 PMC pass of this configuration (FETCH_SIZE x2 +
    WRITE_SIZE, the guide's gfx950 correction), over the kernel time of that pass: the counter-measured GB/s beside
    the algorithmic figure."""
    for rnd in ("r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, pmc_name)
        if os.path.exists(path):
            d = json.load(open(path)).get("_derived", {})
            if "fetch_bytes_per_launch_corrected_x2" in d and d.get("kernel_ms_avg"):
                b = d["fetch_bytes_per_launch_corrected_x2"] + d["write_kib_per_launch"] * 1024.0
                return {"bytes_per_launch": b, "kernel_ms_of_that_pass": d["kernel_ms_avg"], "GBps": b / (d["kernel_ms_avg"] * 1e-3) / 1e9,
                        "frac_of_peak": b / (d["kernel_ms_avg"] * 1e-3) / 1e9 / PEAK_HBM_GBS, "source": os.path.relpath(path, ROOT)}
    return None


def run(label, desc, nx, ny, spp, steps=2, model=None, bound="valu", cpu=None, pmc=None, **kw):
    t0 = time.perf_counter()
    sc = pt.Scene(desc)
    pt.synchronize()
    create_s = time.perf_counter() - t0
    buf = pt.DeviceBuffer(nx * ny * 12)
    sc.render_device(buf.ptr, nx, ny, max(1, spp // 16), 25, **kw)      # warm-up
    pt.synchronize()
    ms = []
    for _ in range(steps):
        sc.render_device(buf.ptr, nx, ny, spp, 25, **kw)
        pt.synchronize()
        ms.append(sc.stats().kernel_ms)
    st = sc.stats()
    k = sum(ms) / len(ms)                          # the mean of the timed launches, as bench.py reports its headline
    out = {"config": label, "nx": nx, "ny": ny, "spp": spp, "kernel_ms": k, "kernel_ms_all": ms, "Mpaths_per_s": nx * ny * spp / k / 1e3,
           "scene_create_s": create_s, "lbvh_nodes": int(pt.library().pt_scene_lbvh_nodes(sc._h, None, 0)),
           "schedule": st.schedule, "pipeline": st.pipeline, "launches": st.n_launches, "shapes": desc.c.n_shapes, "world": desc.c.n_world,
           "lbvh_form": sc.lbvh_form(),
           # a search through the 4-wide form is priced by its own replay (64-byte wide nodes), see tests/make_work_model.py
           "roofline": roofline(model + "_wide" if model and "_lbvh" in model and sc.lbvh_form() == 4 else model, bound,
                                nx * ny * spp / (k * 1e-3))}
    if out["roofline"] is not None and pmc:
        out["roofline"]["measured_hbm"] = measured_hbm(pmc, k)
    if cpu is not None:
        out["cpu_baseline"] = cpu_baseline(desc, *cpu)
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--grid", type=int, default=1000, help="height-field resolution (1000 -> 2 000 000 triangles)")
    ap.add_argument("--only", default="")
    ap.add_argument("--pipeline", choices=("fused", "wavefront"), default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    q = 16 if a.quick else 1
    pt.set_device(0)
    kw = {"pipeline": abi.PIPELINE_WAVEFRONT} if a.pipeline == "wavefront" else {}
    res = []
    cpu = (lambda *sample: None) if (a.no_cpu_baseline or a.quick or a.pipeline == "wavefront") else (lambda *sample: sample)
    if a.only in ("", "c1"):
        res.append(run("C1 random_scene 1920x1080 LBVH", pt.builtin_scene("random", 16 / 9, accel=abi.ACCEL_LBVH), 1920, 1080, 1024 // q, model="random_lbvh",
                       cpu=cpu(1920, 1080, 2, "the same scene over the flat list in reference order"), pmc="pmc_c1_rtiow_lbvh_walk.json", **kw))
    if a.only in ("", "c1list"):
        res.append(run("C1 random_scene 1920x1080 flat list (reference order)", pt.builtin_scene("random", 16 / 9), 1920, 1080, 64 // q if q < 16 else 4, steps=1, model="random", **kw))
    if a.only in ("", "c3"):
        res.append(run("C3 final 1920x1080, synthetic 512x256 earth map, reference BVH walked as is",
                       pt.builtin_scene("final", 16 / 9, earth_rgb=synthetic_earthmap()), 1920, 1080, 256 // q, model="final",
                       cpu=cpu(1920, 1080, 16, "the same scene"), pmc="pmc_c3_final_list.json", **kw))
    if a.only in ("", "c3lbvh"):
        res.append(run("C3 final 1920x1080, synthetic 512x256 earth map, boxes of the reference BVH in the library's LBVH",
                       pt.builtin_scene("final", 16 / 9, earth_rgb=synthetic_earthmap(), accel=abi.ACCEL_LBVH), 1920, 1080, 256 // q, model="final_lbvh",
                       cpu=cpu(1920, 1080, 16, "the same scene (the checker walks the reference's own BVH)"), pmc="pmc_c3_final_lbvh.json", **kw))
    if a.only in ("", "c3noearth"):
        res.append(run("C3 final 1920x1080 without the earth sphere (the reference's device branch), LBVH",
                       pt.builtin_scene("final", 16 / 9, accel=abi.ACCEL_LBVH), 1920, 1080, 256 // q, model="final_lbvh_no_earth", **kw))
    if a.only in ("", "c4"):
        d = pt.trimesh_scene(a.grid, 16 / 9)
        d.c.accel = abi.ACCEL_LBVH
        res.append(run("C4 height field %d triangles in the Cornell enclosure 3840x2160 LBVH" % (2 * a.grid * a.grid), d, 3840, 2160, 64 // q, steps=2, model="trimesh_lbvh", bound="hbm",
                       cpu=cpu(32, 18, 1, "the same scene as a flat list of 2 000 006 shapes (the reference's only way to hold triangles), a 32x18 frame"),
                       pmc="pmc_c4_trimesh_walk.json", **kw))
    return res


if __name__ == "__main__":
    main()
