"""pt_render_device_noalloc: the render entry point that allocates no device memory (its pixel-slot
counter lives in a static device array of the library, one copy per GPU).  Its frames must equal
pt_render_device's bit for bit, call after call, a smaller frame after a larger one included: the
counter keeps the previous call's final value and has to be cleared by every call."""
import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def both(sc, nx, ny, ns, **kw):
    frame = pt.DeviceBuffer(nx * ny * 12)
    sc.render_device(frame.ptr, nx, ny, ns, **kw)
    pt.synchronize()
    want = frame.to_host().copy()
    frame.from_host(np.full(nx * ny * 3, 7.0, np.float32))            # the call must write every value itself
    sc.render_device_noalloc(frame.ptr, nx, ny, ns, **kw)
    pt.synchronize()
    got = frame.to_host().copy()
    return got.reshape(ny, nx, 3), want.reshape(ny, nx, 3)


CASES = [("cornell", 1.0, None, abi.SCHEDULE_REFILL), ("cornell", 1.0, None, abi.SCHEDULE_AUTO), ("random", 2.0, abi.ACCEL_LBVH, abi.SCHEDULE_REFILL),
         ("final", 16 / 9, abi.ACCEL_LBVH, abi.SCHEDULE_AUTO), ("trimesh", 1.0, abi.ACCEL_LBVH, abi.SCHEDULE_WALK)]


@pytest.mark.parametrize("name,aspect,accel,schedule", CASES)
def test_noalloc_render_equals_render_device_twice_in_a_row(oracle, name, aspect, accel, schedule):
    if name == "trimesh":
        d = pt.trimesh_scene(16, aspect)                        # 512 triangles in the Cornell enclosure
        d.c.accel = accel
    else:
        d = pt.builtin_scene(name, aspect, accel=accel)
    sc = pt.Scene(d)
    # a larger frame first, then a smaller one: a counter left at the first call's total would end the second at once
    for nx, ny, ns in ((200, 120, 3), (64, 40, 4), (8, 8, 1)):
        got, want = both(sc, nx, ny, ns, schedule=schedule)
        assert np.array_equal(bits(got), bits(want)), "%s %dx%dx%d schedule %d" % (name, nx, ny, ns, schedule)
        assert want.max() > 0
    # and against the oracle, sharded
    got, want = both(sc, 64, 40, 4, schedule=schedule, shard_index=1, shard_count=3)
    ref = oracle.render(d, 64, 40, 4, shard_index=1, shard_count=3)
    assert np.array_equal(bits(got), bits(ref)) and np.array_equal(bits(want), bits(ref))
    st = sc.stats()
    assert st.paths > 0 and st.kernel_ms > 0


def test_noalloc_render_across_scenes_shares_one_counter():
    """Two scenes on the same GPU use the same static counter: called one after the other they do not disturb each other."""
    a = pt.Scene(pt.builtin_scene("cornell", 1.0))
    b = pt.Scene(pt.builtin_scene("random", 2.0, accel=abi.ACCEL_LBVH))
    for sc in (a, b, a, b):
        got, want = both(sc, 96, 48, 2, schedule=abi.SCHEDULE_REFILL)
        assert np.array_equal(bits(got), bits(want))


def test_noalloc_render_limits_are_errors():
    sc = pt.Scene(pt.builtin_scene("cornell", 1.0))
    frame = pt.DeviceBuffer(64 * 64 * 12)
    with pytest.raises(pt.PtError) as e:
        sc.render_device_noalloc(frame.ptr, 64, 64, 1, pipeline=abi.PIPELINE_WAVEFRONT)
    assert e.value.code == abi.ERR_UNSUPPORTED
    with pytest.raises(pt.PtError) as e:                             # 1x1 tiles at 8192^2: more wave tiles than the counter can number
        sc.render_device_noalloc(frame.ptr, 8192, 8192, 1, tile=(1, 1))
    assert e.value.code == abi.ERR_INVALID and "PT_NOALLOC_MAX_WAVE_TILES" in str(e.value)
    with pytest.raises(pt.PtError):
        sc.render_device_noalloc(frame.ptr, 0, 64, 1)
    sc.render_device_noalloc(frame.ptr, 64, 64, 1)                    # still usable
    pt.synchronize()
    assert np.isfinite(frame.to_host()).all()


def test_noalloc_render_takes_chunks_once_the_scene_holds_their_workspace():
    """The chunked tiles keep state in memory the scene allocates on first use.  A call that may not allocate cannot make
    that memory, but it may use what an earlier pt_render_device call left: then AUTO picks CHUNKS here as it does there
    (the headline frame is a sixth faster in chunks), and the frames are the same bit for bit.  Without the memory,
    asking for CHUNKS is an error and AUTO runs the tiles whole."""
    nx = ny = 512
    sc = pt.Scene(pt.builtin_scene("cornell", 1.0))
    frame = pt.DeviceBuffer(nx * ny * 12)
    with pytest.raises(pt.PtError) as e:
        sc.render_device_noalloc(frame.ptr, nx, ny, 8, schedule=abi.SCHEDULE_CHUNKS, chunk_samples=4)
    assert e.value.code == abi.ERR_UNSUPPORTED and "earlier" in str(e.value)
    sc.render_device_noalloc(frame.ptr, nx, ny, 8)
    pt.synchronize()
    assert sc.stats().schedule == abi.SCHEDULE_TILES
    # an ordinary call in chunks leaves the workspace with the scene ...
    sc.render_device(frame.ptr, nx, ny, 8, schedule=abi.SCHEDULE_CHUNKS, chunk_samples=4)
    pt.synchronize()
    want = frame.to_host().copy()
    assert sc.stats().schedule == abi.SCHEDULE_CHUNKS and sc.stats().n_chunks == 2
    # ... and the call that may not allocate runs on it, twice, the second time on a smaller shard
    for kw in (dict(), dict(shard_index=1, shard_count=3)):
        frame.from_host(np.full(nx * ny * 3, 7.0, np.float32))
        sc.render_device_noalloc(frame.ptr, nx, ny, 8, schedule=abi.SCHEDULE_CHUNKS, chunk_samples=4, **kw)
        pt.synchronize()
        got = frame.to_host().copy()
        assert sc.stats().schedule == abi.SCHEDULE_CHUNKS
        if kw:
            sc.render_device(frame.ptr, nx, ny, 8, schedule=abi.SCHEDULE_TILES, **kw)
            pt.synchronize()
            want = frame.to_host().copy()
        assert np.array_equal(bits(got), bits(want))
    # a larger frame than the workspace covers: refused again
    big = pt.DeviceBuffer(1024 * 1024 * 12)
    with pytest.raises(pt.PtError):
        sc.render_device_noalloc(big.ptr, 1024, 1024, 8, schedule=abi.SCHEDULE_CHUNKS, chunk_samples=4)


def _hip_runtime():
    """The HIP runtime libptcore.so itself is linked against (already mapped: no second runtime is loaded)."""
    import ctypes
    from gpupathtracer_amd.hostgroup import mapped_runtimes
    pt.library()
    hip, _ = mapped_runtimes()
    assert len(hip) == 1
    return ctypes.CDLL(hip[0])


def test_noalloc_render_captured_into_a_graph_and_replayed():
    """What the entry point is for: a render recorded under HIP graph capture (no allocation may happen
    there), replayed twice, the second time after the frame was overwritten; then the scene renders and
    reports statistics normally again.  The library keeps its timing events out of the captured stream."""
    import ctypes as C
    hip = _hip_runtime()
    sc = pt.Scene(pt.builtin_scene("random", 2.0, accel=abi.ACCEL_LBVH))
    nx, ny, ns = 160, 80, 3
    frame = pt.DeviceBuffer(nx * ny * 12)
    sc.render_device(frame.ptr, nx, ny, ns)
    pt.synchronize()
    want = frame.to_host().copy()
    stats_before = sc.stats()
    stream, graph, exe = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipStreamBeginCapture(stream, 0) == 0                  # hipStreamCaptureModeGlobal
    try:
        sc.render_device_noalloc(frame.ptr, nx, ny, ns, stream=stream.value)
        with pytest.raises(pt.PtError):                               # the staged pipeline cannot be recorded
            sc.render_device(frame.ptr, nx, ny, ns, stream=stream.value, pipeline=abi.PIPELINE_WAVEFRONT)
    finally:
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    assert hip.hipGraphInstantiate(C.byref(exe), graph, None, None, 0) == 0
    for _ in range(2):
        frame.from_host(np.full(nx * ny * 3, 7.0, np.float32))
        assert hip.hipGraphLaunch(exe, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        assert np.array_equal(bits(frame.to_host()), bits(want))
    assert sc.stats().kernel_ms == stats_before.kernel_ms             # still the render that ran outside the graph
    frame.from_host(np.full(nx * ny * 3, 7.0, np.float32))
    sc.render_device_noalloc(frame.ptr, nx, ny, ns)                   # the scene's own events were never captured
    pt.synchronize()
    assert np.array_equal(bits(frame.to_host()), bits(want))
    assert sc.stats().paths == nx * ny * ns
    assert hip.hipGraphExecDestroy(exe) == 0 and hip.hipGraphDestroy(graph) == 0 and hip.hipStreamDestroy(stream) == 0
