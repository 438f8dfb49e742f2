"""The N>1 path on CPU: two gloo ranks each trace their tile shard with the oracle, the frames are
summed onto rank 0, and the result must equal the unsharded frame bit for bit.  This is the host
logic of bench.py's multi-GPU leg (shard ownership + reduce) without a GPU."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, store_path, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import gpupathtracer_amd as pt
    from tests import _oracle
    dist.init_process_group("gloo", init_method="file://" + store_path, rank=rank, world_size=world)
    orc = _oracle.load()
    d = pt.builtin_scene("cornell", 72 / 40)
    part = orc.render(d, 72, 40, 3, shard_index=rank, shard_count=world, tile=(8, 8), threads=2)
    t = torch.from_numpy(part.copy())
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    clock = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(clock, op=dist.ReduceOp.MAX)
    assert float(clock.item()) == float(world)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_stitch_the_frame(world):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import gpupathtracer_amd as pt
    from tests import _oracle
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "frame.npy")
        mp.spawn(_worker, args=(world, store, out), nprocs=world, join=True)
        got = np.load(out)
    d = pt.builtin_scene("cornell", 72 / 40)
    full = _oracle.load().render(d, 72, 40, 3)
    np.testing.assert_array_equal(got.view(np.uint32), full.view(np.uint32))


def test_every_pixel_has_exactly_one_owner():
    sys.path.insert(0, ROOT)
    import gpupathtracer_amd as pt
    from tests import _oracle
    orc = _oracle.load()
    d = pt.builtin_scene("simple_spheres", 50 / 30)      # sky everywhere: no pixel renders to exact zero
    for world, tile in ((4, (8, 8)), (7, (16, 8)), (8, (24, 40))):
        owners = np.zeros((30, 50), np.int32)
        for r in range(world):
            owners += (orc.render(d, 50, 30, 1, shard_index=r, shard_count=world, tile=tile).sum(axis=2) > 0)
        assert (owners == 1).all()


def _file_group_worker(rank, world, directory, out_path):
    sys.path.insert(0, ROOT)
    import gpupathtracer_amd as pt
    from gpupathtracer_amd.hostgroup import FileGroup
    from tests import _oracle
    g = FileGroup(rank, world, directory=directory, timeout=120)
    uid = g.broadcast("rccl_unique_id", bytes(range(128)) if rank == 0 else None)     # what bench.py hands round
    assert uid == bytes(range(128))
    assert g.allreduce_min_int(1 if rank != 1 else 0) == 0                           # one rank without a communicator: all fall back
    assert g.allreduce_min_int(1) == 1
    orc = _oracle.load()
    d = pt.builtin_scene("cornell", 72 / 40)
    for step in range(3):                                                             # bench.py's loop: render shard, sum on rank 0
        part = orc.render(d, 72, 40, 2 + step, shard_index=rank, shard_count=world, tile=(8, 8), threads=2)
        total = g.reduce_sum(part, root=0)
        assert (total is None) == (rank != 0)
        if rank == 0:
            np.save(out_path % step, total)
    g.barrier()
    assert g.allreduce_max(10.0 + rank) == 10.0 + world - 1
    assert len([f for f in os.listdir(directory) if f.startswith("sum.")]) <= world  # earlier rounds' frames are gone
    g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_file_group_drives_the_bench_logic_without_any_gpu_runtime(world):
    """bench.py's N > 1 host logic over its own rendezvous code (gpupathtracer_amd/hostgroup.py), as
    plain processes: id broadcast, agreement on the fallback, shard frames summed on rank 0 equal to
    the unsharded frame bit for bit, max-over-ranks clock, clean-up."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    import gpupathtracer_amd as pt
    from tests import _oracle
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as tmp:
        directory, out = os.path.join(tmp, "group"), os.path.join(tmp, "frame%d.npy")
        procs = [ctx.Process(target=_file_group_worker, args=(r, world, directory, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert not os.path.exists(directory)                                          # rank 0 removed it after the last rank left
        d = pt.builtin_scene("cornell", 72 / 40)
        for step in range(3):
            full = _oracle.load().render(d, 72, 40, 2 + step)
            np.testing.assert_array_equal(np.load(out % step).view(np.uint32), full.view(np.uint32))


def test_mapped_runtime_report_names_one_hip_runtime():
    """A process that loads the library and nothing else maps exactly one HIP runtime, the one the
    library links; bench.py prints this report and refuses to pass with two."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import gpupathtracer_amd as pt; pt.library(); "
            "from gpupathtracer_amd.hostgroup import mapped_runtimes; print(json.dumps([mapped_runtimes(), 'torch' in sys.modules]))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, check=True).stdout
    (hip, rccl), torch_loaded = __import__("json").loads(out.strip().splitlines()[-1])
    assert len(hip) == 1 and len(rccl) == 1 and not torch_loaded
    assert "/torch/" not in hip[0] and "/torch/" not in rccl[0]


def test_ranks_named_by_the_launcher_need_no_common_parent(monkeypatch, tmp_path):
    """A launcher that gives its ranks different parents names the launch instead (PT_RENDEZVOUS_ID, or a
    TORCHELASTIC_RUN_ID other than the default "none"): the directory then does not depend on the parent."""
    from gpupathtracer_amd import hostgroup
    monkeypatch.setenv("PT_RENDEZVOUS_ROOT", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29517")
    monkeypatch.delenv("PT_RENDEZVOUS_ID", raising=False)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    by_parent = hostgroup.FileGroup(0, 2).dir
    assert ("ppid_%d_" % os.getppid()) in by_parent
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job 42/a")
    by_run = hostgroup.FileGroup(0, 2).dir
    assert by_run.endswith("id_job_42_a") and str(os.getppid()) not in os.path.basename(by_run).split("id_")[1]
    monkeypatch.setenv("PT_RENDEZVOUS_ID", "mine")                       # the library's own variable wins
    assert hostgroup.FileGroup(1, 2).dir.endswith("29517_id_mine")
    assert oct(os.stat(by_run).st_mode & 0o777) == oct(0o700)


def test_a_rank_left_alone_fails_soon_and_names_the_directory(monkeypatch, tmp_path):
    from gpupathtracer_amd import hostgroup
    monkeypatch.setenv("PT_RENDEZVOUS_ROOT", str(tmp_path))
    monkeypatch.setenv("PT_RENDEZVOUS_TIMEOUT", "0.3")
    g = hostgroup.FileGroup(1, 2)
    with pytest.raises(TimeoutError) as e:
        g.broadcast("rccl_unique_id")
    assert g.dir in str(e.value) and "PT_RENDEZVOUS_ID" in str(e.value)


def test_a_rendezvous_directory_others_can_write_is_refused(tmp_path):
    from gpupathtracer_amd import hostgroup
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    with pytest.raises(PermissionError):
        hostgroup.FileGroup(0, 2, directory=str(loose))


def test_a_file_left_by_an_old_launch_is_waited_past(tmp_path):
    from gpupathtracer_amd import hostgroup
    d = tmp_path / "g"
    g = hostgroup.FileGroup(1, 2, directory=str(d), timeout=0.3)
    stale = d / "rccl_unique_id"
    stale.write_bytes(b"old")
    os.utime(stale, (1.0e9, 1.0e9))
    with pytest.raises(TimeoutError):
        g.broadcast("rccl_unique_id")
    hostgroup.FileGroup(0, 2, directory=str(d)).broadcast("rccl_unique_id", b"new")
    assert g.broadcast("rccl_unique_id") == b"new"
