#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s, whole job, Cornell box 1024x1024 at 4096 spp, tile-sharded over
N MI355X of one node.

One process per GPU (the driver starts N of them through torch.distributed.run).  A "step" is one
complete render of the frame: every rank traces its 8x8-tile shard of the image with the scene
already resident in HBM, then the shards are summed onto rank 0 with one RCCL reduce over xGMI.
A path is one camera sample (one color() call of the reference, src/ptMain.cu:161).

The ranks load no GPU runtime but the one libptcore.so links: RCCL's unique id travels from rank 0
to the others through a directory on the local file system (gpupathtracer_amd/hostgroup.py), and
barriers and the max-over-ranks clock go through the library's own RCCL communicator.  N > 1 thus
runs on the same HIP and RCCL as N = 1; the line reports which runtime files are mapped and the
run fails if two HIP runtimes are.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic work of one path under the reference's arithmetic (one FLOP = one fp32
# add/sub/mul/compare/min/max/div/sqrt, no FMA credit because parity forbids contraction), counted
# by the instrumented build of the CPU checker: tests/make_work_model.py -> bench/work_model.json.
# Cornell: 52.4 rectangle tests, 2.81 closest-hit searches, 1.81 diffuse scatters, 799 FLOP per path.
with open(os.path.join(ROOT, "bench", "work_model.json")) as _f:
    FLOP_PER_PATH = {k: v["flop_per_path"] for k, v in json.load(_f)["configs"].items()}
# HBM bytes one launch moves, from the committed rocprofv3 PMC pass (FETCH_SIZE + WRITE_SIZE with
# the guide's unit and gfx950 corrections): the frame is written once, the scene stays in cache.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03", "pmc_cornell1024x4096_k_render_chunks.json")
# fp32 vector peak without FMA: 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz (with FMA: 157.3 TFLOP/s).  Measured
# (tools/valu_probe.hip): plain add/mul/fma on vector registers issue at that rate; compares, selects,
# min/max, integer multiplies, fp64, conversions and the scale / fix-up steps of a divide at half of it.
PEAK_VALU_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def library_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--depth", type=int, default=25)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--cpu-spp", type=int, default=512, help="samples per pixel of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-staged", action="store_true", help="skip the one render on the staged (wavefront) pipeline reported beside the headline")
    ap.add_argument("--frame-hash", action="store_true", help="add the SHA-1 of the final frame on rank 0 to the result line")
    ap.add_argument("--shard", choices=("tiles", "samples", "hybrid"), default="tiles",
                    help="tiles: whole pixels per rank, the reference's image bit for bit (default).  samples: every rank "
                         "traces spp/N samples of every pixel from its own seeds -- not the reference's random sequence, "
                         "equal to it up to sampling noise.  hybrid: --tile-shards T tile shards x N/T sample batches, "
                         "the smallest departure from the reference's sequence that keeps every GPU busy")
    ap.add_argument("--tile-shards", type=int, default=2, help="tile shards of the hybrid sharding (the ranks divide into that many)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import gpupathtracer_amd as pt
    from gpupathtracer_amd import hostgroup

    n_dev = pt.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible")
    pt.set_device(local_rank % n_dev)

    nx = ny = args.size
    desc = pt.builtin_scene(args.scene, float(nx) / float(ny))
    scene = pt.Scene(desc)
    n_floats = 3 * nx * ny
    frame = pt.DeviceBuffer(n_floats * 4)

    # N > 1: rank 0's RCCL id reaches the others through files; everything after that is RCCL.
    # PT_BENCH_COLLECTIVE=files keeps barriers, clock and the frame sum on files too: a rehearsal of
    # this script's logic on a box whose ranks share one GPU, where RCCL cannot connect them.
    collective = "none"
    group = comm = None
    if world > 1:
        group = hostgroup.FileGroup(rank, world)
        uid = group.broadcast("rccl_unique_id", pt.Comm.unique_id() if rank == 0 else None)
        want_rccl = os.environ.get("PT_BENCH_COLLECTIVE", "rccl") == "rccl"
        if want_rccl:
            try:
                comm = pt.Comm(uid, rank, world)
            except pt.PtError as e:      # keep the run alive, but say so in the result line
                sys.stderr.write("[bench] rank %d: RCCL communicator failed (%s)\n" % (rank, e))
        if group.allreduce_min_int(1 if comm is not None else 0) == 0:
            if comm is not None:
                comm.close()
                comm = None
            collective = "files-host-reduce" + ("" if not want_rccl else " (RCCL failed)")
        else:
            collective = "rccl-reduce"

    kernel_ms = []

    # how the ranks divide the frame: (tile shards, sample batches).  (N, 1) is the reference's image bit for bit.
    T = args.tile_shards
    hybrid_ok = world >= 4 and T >= 2 and world % T == 0 and world // T >= 2 and args.spp % (world // T) == 0
    samples_ok = world > 1 and args.spp % world == 0
    mode = args.shard if world > 1 else "tiles"
    if mode == "samples" and not samples_ok:
        raise SystemExit("bench.py: --shard samples needs spp divisible by the number of GPUs")
    if mode == "hybrid" and not hybrid_ok:
        raise SystemExit("bench.py: --shard hybrid needs at least 4 GPUs, divisible by --tile-shards, and spp divisible by the batches")
    by_samples = mode != "tiles"
    split = {"tiles": (world, 1), "samples": (1, world), "hybrid": (T, max(1, world // T))}

    def step(record, samples):
        """samples: False, or the name of a sharding that divides a pixel's samples ("samples", "hybrid")."""
        if samples:           # raw sums of this rank's batch of its tile shard; added by the reduce, averaged + gamma'd once on the root
            tiles, batches = split[samples]
            scene.render_device(frame.ptr, nx, ny, args.spp // batches, args.depth, shard_index=rank % tiles, shard_count=tiles, tile=(8, 8),
                                batch_index=rank // tiles, batch_count=batches, raw_sum=1)
        else:
            scene.render_device(frame.ptr, nx, ny, args.spp, args.depth, shard_index=rank, shard_count=world, tile=(8, 8))
        if record:
            kernel_ms.append(scene.stats().kernel_ms)      # HIP events on the launch stream
        if world > 1:
            if comm is not None:
                comm.reduce_frame(frame.ptr, n_floats, root=0)
            else:
                total = group.reduce_sum(frame.to_host(), root=0)
                if rank == 0:
                    frame.from_host(total)
            if samples and rank == 0:
                pt.frame_finish(frame.ptr, n_floats, args.spp)

    def barrier():
        if comm is not None:
            comm.barrier()
        elif group is not None:
            group.barrier()

    def timed(samples, record):
        for _ in range(args.warmup):
            step(False, samples)
        pt.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(record, samples)
        pt.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = comm.allreduce_max(dt)
        elif group is not None:
            dt = group.allreduce_max(dt)
        return dt

    dt = timed(mode if by_samples else False, True)
    fused_stats = scene.stats()              # of the headline's last launch, before anything else renders
    frame_sha1 = None
    if args.frame_hash and rank == 0:       # the headline's frame, before the second sharding below overwrites it
        import hashlib
        frame_sha1 = hashlib.sha1(frame.to_host().tobytes()).hexdigest()
    # Beside the headline (tile shards: the reference's image bit for bit), the same frame with the
    # samples of every pixel split over the ranks: the other sharding the task names, equal to the
    # reference's image up to sampling noise only.  Reported, never used for `value`.
    dt_samples = dt_hybrid = None
    noise = None
    if world > 1 and not by_samples:
        one_stream = frame.to_host().reshape(ny, nx, 3).copy() if rank == 0 else None
        batch_frames = {}
        if samples_ok:
            dt_samples = timed("samples", False)
            if rank == 0:
                batch_frames["samples"] = frame.to_host().reshape(ny, nx, 3).copy()
        if hybrid_ok:
            dt_hybrid = timed("hybrid", False)
            if rank == 0:
                batch_frames["hybrid"] = frame.to_host().reshape(ny, nx, 3).copy()
        if batch_frames or samples_ok or hybrid_ok:
            # The tolerance the batched frames are held to: per channel, their RMSE against the one-stream frame within
            # 1.5 x the RMSE of two one-stream frames from disjoint seeds (two independent estimates of the same pixels,
            # sqrt(2) sigma apart however the samples were drawn).  The second one-stream frame: every pixel seeded with
            # i + nx*ny instead of i, tile-sharded like the headline.
            scene.render_device(frame.ptr, nx, ny, args.spp, args.depth, shard_index=rank, shard_count=world, tile=(8, 8), batch_index=1, batch_count=2)
            if comm is not None:
                comm.reduce_frame(frame.ptr, n_floats, root=0)
            else:
                total = group.reduce_sum(frame.to_host(), root=0)
                if rank == 0:
                    frame.from_host(total)
            pt.synchronize()
            if rank == 0:
                other = frame.to_host().reshape(ny, nx, 3)
                rmse = lambda a, b: np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean(axis=(0, 1)))
                floor = rmse(one_stream, other)
                noise = {"rmse_two_one_stream_frames_rgb": [float(v) for v in floor], "tolerance": "RMSE(batched, one-stream) <= 1.5 x that, per channel"}
                for k, f in batch_frames.items():
                    r = rmse(f, one_stream)
                    noise[k] = {"rmse_vs_one_stream_rgb": [float(v) for v in r], "ratio_rgb": [float(v) for v in r / np.maximum(floor, 1e-30)],
                                "within_tolerance": bool((r <= 1.5 * floor).all())}

    # Beside the headline (the fused pipeline), one render of the same frame on the staged pipeline
    # (generate / extend / shade by material queue / accumulate, pt_wavefront.hip): its rate and whether
    # its frame equals the fused one bit for bit.  Reported, never used for `value`.
    staged = None
    if world == 1 and not args.no_staged:
        from gpupathtracer_amd import abi
        fused_frame = frame.to_host().copy()
        scene.render_device(frame.ptr, nx, ny, args.spp, args.depth, pipeline=abi.PIPELINE_WAVEFRONT)
        pt.synchronize()
        st = scene.stats()
        staged = {"value": st.paths / st.kernel_ms / 1e3, "unit": "Mpaths/s", "kernel_ms": st.kernel_ms, "launches": st.n_launches,
                  "frame_equals_fused_bit_for_bit": bool(np.array_equal(frame.to_host().view(np.uint32), fused_frame.view(np.uint32))),
                  "note": "PT_PIPELINE_WAVEFRONT, one launch of k_wf_groups; not part of `value`"}
        # what its stages did, from the counting form of the same kernel (an eighth of the samples: the counters cost it 6 %)
        scene.render_device(frame.ptr, nx, ny, max(1, args.spp // 8), args.depth, pipeline=abi.PIPELINE_WAVEFRONT, stage_stats=1)
        pt.synchronize()
        ss, k_ms = scene.stage_stats(), scene.stats().kernel_ms
        staged["stages"] = {"spp": max(1, args.spp // 8), "kernel_ms": k_ms, "groups": ss["groups"], "rounds_per_group": ss["rounds"] / max(1, ss["groups"]),
                            "extend": {"lanes": ss["extend_lanes"], "queue_occupancy": ss["extend_occupancy"], "time_share": ss["extend_time_share"],
                                       "ms": k_ms * ss["extend_time_share"]},
                            "shade": {"lanes_by_queue": dict(zip(("lambertian", "metal", "dielectric", "isotropic"), ss["shade_lanes"])),
                                      "queue_occupancy": ss["shade_occupancy"], "time_share": 1.0 - ss["extend_time_share"],
                                      "ms": k_ms * (1.0 - ss["extend_time_share"])},
                            "note": "pt_render_stage_stats; occupancy = queue entries / (64 x waves run); time shares from the shader clock in one group of 64"}

    paths_per_step = nx * ny * args.spp
    value = paths_per_step * args.steps / dt / 1e6
    result = None
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        shard_paths = fused_stats.paths              # paths this rank's launch traced
        flop = FLOP_PER_PATH.get(args.scene)
        roofline = None
        if flop is not None:
            achieved = shard_paths * flop / (k_ms * 1e-3) / 1e12
            # HBM bytes per launch come from a committed rocprofv3 PMC pass (counters cannot be read inside
            # this process); the file names the build it profiled, and a number from another build is not reported
            traffic, traffic_note = None, "no PMC pass for this workload"
            if args.scene == "cornell" and nx == 1024 and ny == 1024 and args.spp == 4096 and world == 1 and os.path.exists(PMC_TRAFFIC_FILE):
                with open(PMC_TRAFFIC_FILE) as f:
                    pmc = json.load(f)
                if pmc.get("_libptcore_sha256") == library_sha256(pt.library_path()):
                    d = pmc["_derived"]
                    traffic = d["fetch_bytes_per_launch_corrected_x2"] + d["write_kib_per_launch"] * 1024.0
                    traffic_note = "FETCH_SIZE x2 + WRITE_SIZE per launch, %s (%s)" % (os.path.relpath(PMC_TRAFFIC_FILE, ROOT), pmc.get("_command", ""))
                else:
                    traffic_note = "the committed PMC pass profiled another build of libptcore.so; not reported"
            roofline = {"bound": "valu", "achieved": achieved, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_VALU_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                        "kernel": {2: "k_render_refill", 3: "k_render_walk", 4: "k_render_chunks"}.get(fused_stats.schedule, "k_render_tiles"),
                        "kernel_ms": k_ms, "flop_per_path": flop, "paths_per_launch": shard_paths,
                        "peak_note": "fp32 VALU, one add/mul per lane per cycle, no FMA credit (parity forbids contraction); compares, "
                                     "selects, min/max, fp64 and divide scale/fix-up steps issue at half that rate (%.1f TFLOP/s)"
                                     % (PEAK_VALU_TFLOPS / 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from tests import _oracle
            orc = _oracle.load()
            # every hardware thread this job may use: the box hands a one-GPU job a share of its CPUs and says
            # so in OMP_NUM_THREADS (os.cpu_count() reports the whole machine there)
            visible = os.cpu_count() or 1
            share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            threads = max(1, min(visible, share) if share > 0 else visible)
            c0 = time.perf_counter()
            orc.render(desc, nx, ny, args.cpu_spp, args.depth, threads=threads)
            cdt = time.perf_counter() - c0
            cpu = {"value": nx * ny * args.cpu_spp / cdt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
                   "host_threads_visible": visible, "host_thread_share": share if share > 0 else None,
                   "sample": "%s %dx%d at %d spp, depth %d, OpenMP rows over %d threads (all this job may use; nproc says %d)" %
                             (args.scene, nx, ny, args.cpu_spp, args.depth, threads, visible)}
        result = {
            "metric": "Mpaths/s whole node, Cornell 1024^2 4096spp",
            "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s_box %dx%d %dspp depth%d (BASELINE configs[2])" % (args.scene, nx, ny, args.spp, args.depth)
                       if args.scene == "cornell" else "%s %dx%d %dspp depth%d" % (args.scene, nx, ny, args.spp, args.depth),
                       "sharding": ("%d tile shards x %d sample batches (own seeds per batch: not the reference's sequence)" % split[mode] if by_samples
                                    else "8x8 pixel tiles round-robin over ranks"),
                       "pipeline": "fused", "collective": collective},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        hip_libs, rccl_libs = hostgroup.mapped_runtimes()
        result["runtime"] = {"hip": hip_libs, "rccl": rccl_libs, "torch_loaded": "torch" in sys.modules}
        if frame_sha1 is not None:
            result["frame_sha1"] = frame_sha1
        if staged is not None:
            result["staged_pipeline"] = staged
        if dt_samples is not None:
            result["sample_batch_sharding"] = {
                "value": paths_per_step * args.steps / dt_samples / 1e6, "unit": "Mpaths/s", "ms_per_step": dt_samples / args.steps * 1e3,
                "note": "every rank traces spp/N samples of every pixel from its own seeds, raw sums reduced, finished on rank 0; "
                        "not the reference's random sequence (equal to it within the stated RMSE tolerance), not part of `value`"}
        if dt_hybrid is not None:
            result["hybrid_sharding"] = {
                "value": paths_per_step * args.steps / dt_hybrid / 1e6, "unit": "Mpaths/s", "ms_per_step": dt_hybrid / args.steps * 1e3,
                "tile_shards": split["hybrid"][0], "sample_batches": split["hybrid"][1],
                "note": "rank r traces tile shard r % T of sample batch r // T; batch 0 is the reference's stream, the others their own "
                        "seeds; not part of `value`"}
        if noise is not None:
            result["batched_frames_vs_sampling_noise"] = noise
    hip_libs, _ = hostgroup.mapped_runtimes()
    barrier()
    if comm is not None:
        comm.close()
    if group is not None:
        group.close()
    if result is not None:
        print(json.dumps(result), flush=True)
    if len(hip_libs) > 1:        # two HIP runtimes in one process: whatever was measured, it was not the N = 1 configuration
        raise SystemExit("bench.py: rank %d has more than one HIP runtime mapped: %s" % (rank, hip_libs))


if __name__ == "__main__":
    main()
