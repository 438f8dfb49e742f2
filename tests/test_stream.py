"""The reference's scene byte stream (src/ptStream.cu, every serialize/deserialize member) read into
and written from the flat scene description.  CPU only: host code of the library plus the oracle.

Pins: the survey of the reference recorded the size of world->serialize for random_scene
(27 276 bytes) and cornell_box (1 024 bytes); the writer must produce exactly those."""
import struct

import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

SCENES = [("cornell", 1.0), ("random", 2.0), ("simple_spheres", 2.0), ("simple_light", 2.0), ("cornell_spheres", 1.0)]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def fourcc(s):
    return struct.unpack("<i", s.encode())[0]


def i32(*v):
    return struct.pack("<%di" % len(v), *v)


def f32(*v):
    return struct.pack("<%df" % len(v), *v)


def lamb(r, g, b):
    return i32(fourcc("LAMB")) + i32(fourcc("CNST")) + f32(r, g, b)


CAMERA = f32(*([0.0, 0.0, 5.0] + [-2.0, -1.0, 4.0] + [4.0, 0.0, 0.0] + [0.0, 2.0, 0.0] + [1, 0, 0] + [0, 1, 0] + [0, 0, 1] + [0.0, 1.0, 0.0]))
SKY = i32(fourcc("SKYA"))


def test_world_stream_sizes_match_the_reference_probe():
    for name, aspect, want in (("random", 2.0, 27276), ("cornell", 1.0, 1024)):
        blob, sections = pt.scene_to_stream(pt.builtin_scene(name, aspect))
        assert sections[0] == want
        assert sections[2] == 96                                 # 7 vectors + time0, time1, lens_radius
        assert sum(sections) == len(blob)
    blob, sections = pt.scene_to_stream(pt.builtin_scene("cornell", 1.0))
    assert sections[1] == 28 and sections[3] == 16               # XZRectangle with a null material; CNST + colour
    assert struct.unpack_from("<ii", blob, 0) == (1, 8)          # ListTypeId, count


@pytest.mark.parametrize("name,aspect", SCENES)
def test_round_trip_is_byte_and_image_identical(oracle, name, aspect):
    d = pt.builtin_scene(name, aspect)
    blob, _ = pt.scene_to_stream(d)
    d2 = pt.scene_from_stream(blob)
    blob2, _ = pt.scene_to_stream(d2)
    assert blob2 == blob
    assert d2.c.n_world == d.c.n_world and d2.c.n_lights == d.c.n_lights and d2.c.lights_is_list == d.c.lights_is_list
    assert d2.c.n_materials <= d.c.n_materials                   # per-leaf copies are merged
    a, b = oracle.render(d, 40, 24, 4), oracle.render(d2, 40, 24, 4)
    assert (bits(a) == bits(b)).all()


def test_final_scene_cannot_be_streamed():
    with pytest.raises(pt.PtError) as e:
        pt.scene_to_stream(pt.builtin_scene("final", 16 / 9))
    assert e.value.code == abi.ERR_UNSUPPORTED


def test_hand_written_stream_features(oracle):
    """Nested lists flatten, FlipNormals toggles, a Translate's offset (written after its child) lands
    on every leaf below it, a TriangleMesh record is skipped, a Box with a foreign child expands."""
    sphere = lambda x, r: i32(2) + f32(x, 0.0, 0.0, r) + lamb(0.5, 0.5, 0.5)
    rect = i32(4) + lamb(0.2, 0.3, 0.4) + f32(-1, 1, -1, 1, -2)                    # XYRectangle z=-2
    mesh = i32(14) + i32(2) + b"\0" * 88
    inner = i32(1) + i32(2) + sphere(0.0, 0.5) + i32(7) + i32(7) + i32(7) + rect   # list { sphere, flip^3(rect) }
    odd_box = i32(8) + f32(0, 0, 0, 1, 1, 1) + sphere(3.0, 0.25)                   # Box whose child is a sphere
    world = i32(1) + i32(4) + i32(9) + inner + f32(1.0, 2.0, 3.0) + mesh + odd_box + i32(-1)
    blob = world + i32(-1) + CAMERA + SKY
    desc = pt.scene_from_stream(blob)            # keep the owner alive while reading its arrays
    d = desc.c
    assert d.n_world == 3 and d.n_shapes == 3 and d.n_lights == 0 and d.ambient_type == abi.AMBIENT_SKY
    s0, s1, s2 = d.shapes[0], d.shapes[1], d.shapes[2]
    assert (s0.type, s0.flags, s0.xform_count) == (abi.SHAPE_SPHERE, 0, 1)
    assert (s1.type, s1.flags, s1.xform_count) == (abi.SHAPE_RECT_XY, abi.FLIP_NORMAL, 1)
    assert (s2.type, s2.xform_count) == (abi.SHAPE_SPHERE, 0)
    for s in (s0, s1):
        x = d.xforms[s.xform_first]
        assert x.type == abi.XFORM_TRANSLATE and list(x.v) == [1.0, 2.0, 3.0]
    assert d.n_materials == 2                                    # the two grey spheres share one record
    assert list(d.camera)[:3] == [0.0, 0.0, 5.0]
    # a stream the oracle can render: same picture as the equivalent hand-built world
    img = oracle.render(pt.scene_from_stream(blob), 32, 16, 2)
    assert np.isfinite(img).all() and img.max() > 0


def test_constant_ambient_colour_follows_the_reference_device_side():
    d = pt.builtin_scene("cornell", 1.0)
    blob, sections = pt.scene_to_stream(d)
    red = blob[:-12] + f32(1.0, 0.0, 0.0)
    black = pt.scene_from_stream(red)
    kept = pt.scene_from_stream(red, flags=abi.STREAM_KEEP_AMBIENT_COLOR)
    assert list(black.c.ambient_color) == [0.0, 0.0, 0.0]
    assert list(kept.c.ambient_color) == [1.0, 0.0, 0.0]


@pytest.mark.parametrize("blob,code", [
    (b"", abi.ERR_INVALID),
    (i32(1) + i32(1) + i32(99), abi.ERR_INVALID),                                  # unknown hitable id
    (i32(1) + i32(1) + i32(12), abi.ERR_INVALID),                                  # BVH records are never written
    (i32(1) + i32(1) + i32(2) + f32(0, 0, 0, 1) + i32(fourcc("NOPE")), abi.ERR_INVALID),
    (i32(1) + i32(1) + i32(2) + f32(0, 0, 0, 1) + i32(fourcc("LAMB")) + i32(-1), abi.ERR_INVALID),   # null texture
    (i32(1) + i32(0) + i32(-1) + CAMERA[:50], abi.ERR_INVALID),                    # truncated camera
    (i32(1) + i32(1) + i32(9) + i32(11) + i32(2) + f32(0, 0, 0, 1) + lamb(1, 1, 1) + f32(0.1) +
     i32(fourcc("ISOT")) + i32(fourcc("CNST")) + f32(1, 1, 1) + f32(0, 0, 0), abi.ERR_UNSUPPORTED),  # Translate(ConstantMedium)
])
def test_malformed_streams_are_errors(blob, code):
    with pytest.raises(pt.PtError) as e:
        pt.scene_from_stream(blob)
    assert e.value.code == code


def test_mutated_streams_never_crash():
    """Byte-level damage to valid streams: the reader either refuses the stream or yields a description
    that pt_scene_create's validation can judge (PT_ERR_INVALID, or PT_ERR_NO_DEVICE / success when
    the description is sound) -- never a fault."""
    import ctypes as C
    rng = np.random.default_rng(7)
    lib = pt.library()
    blobs = [pt.scene_to_stream(pt.builtin_scene(n, a))[0] for n, a in (("cornell", 1.0), ("simple_light", 2.0), ("cornell_spheres", 1.0))]
    outcomes = {"refused": 0, "accepted": 0}
    for trial in range(1500):
        b = bytearray(blobs[trial % len(blobs)])
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            if len(b) < 8:
                break
            pos = int(rng.integers(0, len(b)))
            if kind == 0:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos - pos % 4: pos - pos % 4 + 4] = i32(int(rng.integers(-2, 16)))
            else:
                del b[pos:]
        try:
            d = pt.scene_from_stream(bytes(b))
        except pt.PtError as e:
            assert e.code in (abi.ERR_INVALID, abi.ERR_UNSUPPORTED)
            outcomes["refused"] += 1
            continue
        outcomes["accepted"] += 1
        h = C.c_void_p()
        rc = lib.pt_scene_create(d.ptr, C.byref(h))
        assert rc in (abi.OK, abi.ERR_INVALID, abi.ERR_NO_DEVICE, abi.ERR_UNSUPPORTED)
        if rc == abi.OK:
            lib.pt_scene_destroy(h)
    assert outcomes["refused"] > 100 and outcomes["accepted"] > 100


def test_hostile_streams_and_descriptions_are_errors_not_crashes():
    """Inputs built to exhaust the host rather than to be wrong in one field: a chain of CHEK records
    deep enough to overflow the stack by recursion, an IMAG header that announces far more texels
    than the stream holds, and descriptions whose indices point outside their arrays or at themselves."""
    import ctypes as C
    from gpupathtracer_amd import _abi as A
    from tests._desc_tools import OwnedDesc
    sphere_head = i32(1) + i32(1) + i32(2) + f32(0, 0, 0, 1) + i32(fourcc("LAMB"))
    deep = sphere_head + (i32(fourcc("CHEK")) + f32(1.0)) * 200_000
    with pytest.raises(pt.PtError) as e:
        pt.scene_from_stream(deep)
    assert e.value.code == abi.ERR_INVALID and "nested" in str(e.value)
    huge = sphere_head + i32(fourcc("IMAG")) + i32(16384) + i32(16384) + b"\0" * 64     # 768 MiB announced, 64 bytes present
    with pytest.raises(pt.PtError) as e:
        pt.scene_from_stream(huge)
    assert e.value.code == abi.ERR_INVALID and "image" in str(e.value)
    # several images whose offsets would pass 2 GiB are refused before anything is allocated for them
    lib = pt.library()
    size, sections = C.c_size_t(0), (C.c_size_t * 4)()

    def to_stream_rc(g):
        g.finalize()
        return lib.pt_scene_desc_to_stream(g.ptr, None, 0, C.byref(size), sections)

    g = OwnedDesc(pt.builtin_scene("random", 2.0))
    assert to_stream_rc(g) == abi.OK
    chk = next(i for i, t in enumerate(g.lists["textures"]) if t.type == abi.TEX_CHECKER)
    g.lists["textures"][chk].odd = chk                                  # a checker that is its own child
    assert to_stream_rc(g) == abi.ERR_INVALID
    for field, arr, bad in (("material", "shapes", 10_000), ("xform_count", "shapes", 7), ("xform_first", "shapes", -3),
                            ("texture", "materials", 99), ("texture", "materials", 2_000_000_000)):
        g = OwnedDesc(pt.builtin_scene("cornell", 1.0))
        setattr(g.lists[arr][1], field, bad)
        assert to_stream_rc(g) == abi.ERR_INVALID, (field, bad)
    g = OwnedDesc(pt.builtin_scene("trimesh", 1.0))
    tri = next(s for s in g.lists["shapes"] if s.type == abi.SHAPE_TRIANGLE)
    tri.tri_uv = 5_000_000
    assert to_stream_rc(g) == abi.ERR_INVALID
    tex = np.zeros((4, 4, 3), np.uint8)
    g = OwnedDesc(pt.builtin_scene("final", 1.5, earth_rgb=tex))
    img = next(t for t in g.lists["textures"] if t.type == abi.TEX_IMAGE)
    img.nx = 30000
    g.lists["bvhs"], g.lists["bvh_nodes"] = [], []
    g.lists["world"] = [o for o in g.lists["world"] if o.type != abi.OBJ_BVH]
    assert to_stream_rc(g) == abi.ERR_INVALID
