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
