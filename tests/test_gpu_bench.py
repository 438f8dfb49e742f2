"""bench.py itself on the GPU box: the one-GPU line's contract fields, and a two-rank rehearsal of the
multi-GPU path started exactly as the driver starts it (torch.distributed.run), both ranks sharing
the box's one GPU and exchanging frames through files (RCCL cannot connect two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_line_and_two_rank_rehearsal():
    common = ["--steps", "2", "--warmup", "1", "--size", "256", "--spp", "8", "--frame-hash"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--cpu-spp", "2"] + common, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "runtime"):
        assert key in a, key
    assert a["n_gpus"] == 1 and a["steps"] == 2 and a["warmup"] == 1 and a["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(a["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(a["cpu_baseline"])
    assert len(a["runtime"]["hip"]) == 1 and not a["runtime"]["torch_loaded"]
    stages = a["staged_pipeline"]["stages"]                     # what the staged pipeline's stages did (pt_render_stage_stats)
    assert a["staged_pipeline"]["frame_equals_fused_bit_for_bit"]
    assert 0 < stages["extend"]["queue_occupancy"] <= 1 and 0 < stages["shade"]["queue_occupancy"] <= 1
    assert abs(stages["extend"]["time_share"] + stages["shade"]["time_share"] - 1) < 1e-9 and stages["shade"]["lanes_by_queue"]["lambertian"] > 0

    env = dict(os.environ, PT_BENCH_COLLECTIVE="files", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + common, cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["config"]["collective"].startswith("files")
    assert b["frame_sha1"] == a["frame_sha1"]                   # two tile shards summed = the one-GPU frame, bit for bit
    assert b["runtime"]["hip"] == a["runtime"]["hip"]           # the ranks run on the runtime the one-GPU run used
    assert not b["runtime"]["torch_loaded"]
    assert "sample_batch_sharding" in b
    noise = b["batched_frames_vs_sampling_noise"]               # the stated tolerance of the batched frame, measured in the run itself
    assert noise["samples"]["within_tolerance"] and all(0.5 < r < 1.5 for r in noise["samples"]["ratio_rgb"])

    # four ranks: the hybrid of 2 tile shards x 2 sample batches is timed beside the two plain shardings
    four = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), "bench.py", "--gpus", "4"] + common, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert four.returncode == 0, four.stderr[-3000:]
    c = _last_json(four.stdout)
    assert c["n_gpus"] == 4 and c["frame_sha1"] == a["frame_sha1"]
    assert c["hybrid_sharding"]["tile_shards"] == 2 and c["hybrid_sharding"]["sample_batches"] == 2 and c["hybrid_sharding"]["value"] > 0
    assert c["batched_frames_vs_sampling_noise"]["hybrid"]["within_tolerance"] and c["batched_frames_vs_sampling_noise"]["samples"]["within_tolerance"]
