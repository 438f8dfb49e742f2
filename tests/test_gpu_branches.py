"""GPU parity for the branches of the radiance path that no built-in scene reaches: the thin-lens
camera (include/ptCamera.h:28-34 with lens_radius > 0, randomInUnitDisk include/ptRNG.h:145-150), a
camera origin with a zero component, MovingSphere's uv (include/ptSphere.h:22-28) under an image
texture, the uv a ConstantMedium hit leaves untouched (include/ptMedium.h:46-49), and the
2 000 000-triangle configuration at its full scale.  Bar: bit-exact against the CPU oracle.
"""
import ctypes as C

import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import _abi as A
from gpupathtracer_amd import abi
from gpupathtracer_amd.api import RAY_DTYPE
from tests._desc_tools import OwnedDesc, look_at_camera

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bit_exact(a, b, what):
    same = bits(a) == bits(b)
    assert same.all(), "%s: %d of %d values differ" % (what, (~same).sum(), same.size)


def assert_same_hits(oracle, desc, scene, rays, what):
    ids_o, t_o = oracle.trace_closest(desc, rays)
    ids_g, t_g = scene.trace_closest(rays)
    np.testing.assert_array_equal(ids_g, ids_o, err_msg=what)
    np.testing.assert_array_equal(bits(t_g), bits(t_o), err_msg=what)
    return ids_o


LENS_CASES = [("random", 2.0, 80, 40, 4, 0.05, None), ("random", 2.0, 80, 40, 4, 1.0, None),
              ("cornell", 1.0, 64, 64, 8, 0.05, None), ("cornell", 1.0, 64, 64, 8, 1.0, None),
              ("final", 16 / 9, 64, 36, 4, 2.0, None),
              # aperture 0 but an origin component of exactly 0: the pinhole shortcut must not be taken
              ("random", 2.0, 80, 40, 4, 0.0, 2), ("cornell", 1.0, 64, 64, 8, 0.0, 0), ("cornell", 1.0, 64, 64, 8, 0.5, 1)]


@pytest.mark.parametrize("name,aspect,nx,ny,ns,lens_radius,zero_axis", LENS_CASES)
def test_thin_lens_camera_bit_exact(oracle, name, aspect, nx, ny, ns, lens_radius, zero_axis):
    ref = None
    for accel in (abi.ACCEL_LIST, abi.ACCEL_LBVH):
        d = pt.builtin_scene(name, aspect, accel=accel)
        cam = d.c.camera
        cam[23] = lens_radius
        if zero_axis is not None:                       # move the eye onto a coordinate plane, the view with it
            cam[3 + zero_axis] -= cam[zero_axis]
            cam[zero_axis] = 0.0
        sc = pt.Scene(d)
        if ref is None:
            ref = oracle.render(d, nx, ny, ns)
            assert np.isfinite(ref).all() and ref.max() > 0
        what = "%s lens %g zero %s accel %d" % (name, lens_radius, zero_axis, accel)
        for schedule in (abi.SCHEDULE_AUTO, abi.SCHEDULE_TILES, abi.SCHEDULE_REFILL):
            assert_bit_exact(sc.render(nx, ny, ns, schedule=schedule), ref, what + " schedule %d" % schedule)
        assert_bit_exact(sc.render(nx, ny, ns, pipeline=abi.PIPELINE_WAVEFRONT), ref, what + " staged")
        assert_same_hits(oracle, d, sc, oracle.record_rays(d, 32, 32, 6, 30000), what)
    # the lens must matter: the same frame through a pinhole differs
    if lens_radius > 0:
        d0 = pt.builtin_scene(name, aspect)
        assert not np.array_equal(bits(pt.Scene(d0).render(nx, ny, ns)), bits(ref))


def test_thin_lens_through_progressive_passes_and_shards(oracle):
    d = pt.builtin_scene("cornell", 1.0)
    d.c.camera[23] = 0.75
    sc = pt.Scene(d)
    ref = oracle.render(d, 72, 40, 6)
    img, cancelled = sc.render_progressive(72, 40, 6, 4)
    assert not cancelled
    assert_bit_exact(img, ref, "passes")
    acc = np.zeros_like(ref)
    for s in range(3):
        acc += sc.render(72, 40, 6, shard_index=s, shard_count=3)
    assert_bit_exact(acc, ref, "shards")


def _uv_lab(order, accel, moving_y=0.2, boundary="sphere"):
    """A room small enough that a MovingSphere's world-space hit point has |y| < 1 (its uv is then
    finite: get_uv takes asinf of the un-normalised point): an image-textured floor, a lamp, a moving
    sphere, a plain sphere and a smoke ball, all reading one image; `order` permutes the world list,
    which decides whose uv a plain sphere or a medium shows."""
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, (16, 32, 3), dtype=np.uint8)
    base = pt.builtin_scene("final", 16 / 9, earth_rgb=tex)
    g = OwnedDesc(base)
    image = next(i for i, t in enumerate(g.lists["textures"]) if t.type == abi.TEX_IMAGE)
    const = next(i for i, t in enumerate(g.lists["textures"]) if t.type == abi.TEX_CONSTANT)
    checker = A.PtTexture(type=abi.TEX_CHECKER, odd=image, even=const, scale=3.0)
    g.lists["textures"].append(checker)
    chk = len(g.lists["textures"]) - 1
    mats = g.lists["materials"]
    mats.append(A.PtMaterial(type=abi.MAT_LAMBERTIAN, texture=image)); lam_img = len(mats) - 1
    mats.append(A.PtMaterial(type=abi.MAT_LAMBERTIAN, texture=chk)); lam_chk = len(mats) - 1
    mats.append(A.PtMaterial(type=abi.MAT_ISOTROPIC, texture=image)); iso_img = len(mats) - 1
    mats.append(A.PtMaterial(type=abi.MAT_DIFFUSE_LIGHT, texture=image)); light_img = len(mats) - 1

    def shape(kind, p, material, flags=0):
        s = A.PtShape(type=kind, flags=flags, material=material, xform_first=0, xform_count=0, tri_uv=-1)
        for k, v in enumerate(p):
            s.p[k] = float(np.float32(v))
        return s

    shapes = [shape(abi.SHAPE_RECT_XZ, [-3, 3, -3, 3, -0.9], lam_img),                       # 0 floor (writes uv)
              shape(abi.SHAPE_RECT_XZ, [-1, 1, -1, 1, 2.5], light_img, flags=abi.FLIP_NORMAL),  # 1 lamp, emits the image
              shape(abi.SHAPE_MOVING_SPHERE, [0, 0, 0, 0, moving_y, 0, 0, 1, 0.7], lam_img),  # 2 moving sphere: uv from atan2f/asinf
              shape(abi.SHAPE_SPHERE, [1.3, -0.3, 0.4, 0.5], lam_chk),                         # 3 plain sphere: stale uv through a checker of the image
              {"sphere": shape(abi.SHAPE_SPHERE, [-1.3, -0.2, 0.3, 0.6], lam_img),             # 4 boundary of the smoke ball
               "box": shape(abi.SHAPE_BOX, [-1.9, -0.8, -0.3, -0.7, 0.4, 0.9], lam_img),       #   ... or of a smoke box (the reference's cornell_smoke)
               "moving": shape(abi.SHAPE_MOVING_SPHERE, [-1.3, -0.2, 0.3, -1.2, 0.1, 0.3, 0, 1, 0.6], lam_img)}[boundary],
              shape(abi.SHAPE_RECT_XY, [-3, 3, -0.9, 2.5, -2.5], lam_chk),                     # 5 back wall
              shape(abi.SHAPE_RECT_XZ, [-1, 1, -1, 1, 2.5], -1)]                               # 6 the light shape
    g.lists["shapes"] = shapes
    g.lists["xforms"], g.lists["bvh_nodes"], g.lists["bvhs"], g.lists["tri_uvs"] = [], [], [], []
    g.lists["media"] = [A.PtMedium(boundary=4, density=1.5, material=iso_img)]
    entries = {"floor": A.PtObject(type=abi.OBJ_SHAPE, index=0), "lamp": A.PtObject(type=abi.OBJ_SHAPE, index=1),
               "moving": A.PtObject(type=abi.OBJ_SHAPE, index=2), "plain": A.PtObject(type=abi.OBJ_SHAPE, index=3),
               "smoke": A.PtObject(type=abi.OBJ_MEDIUM, index=0), "wall": A.PtObject(type=abi.OBJ_SHAPE, index=5)}
    g.lists["world"] = [entries[k] for k in order]
    g.lists["lights"] = [6]
    g.scalars["lights_is_list"] = 0
    g.scalars["accel"] = accel
    g.camera = look_at_camera((0.3, 0.8, 4.5), (0, 0, 0), (0, 1, 0), 40, 16 / 9, 0.0, 4.5)
    return g.finalize()


UV_ORDERS = [("floor", "lamp", "moving", "plain", "smoke", "wall"), ("smoke", "plain", "moving", "wall", "lamp", "floor"),
             ("wall", "floor", "plain", "smoke", "moving", "lamp"), ("moving", "smoke", "floor", "wall", "plain", "lamp")]


@pytest.mark.parametrize("order", UV_ORDERS)
def test_moving_sphere_and_medium_uv_under_an_image_texture(oracle, order):
    ref = None
    for accel in (abi.ACCEL_LIST, abi.ACCEL_LBVH):
        g = _uv_lab(order, accel)
        sc = pt.Scene(g)
        if ref is None:
            ref = oracle.render(g, 96, 54, 6)
            rays = oracle.record_rays(g, 48, 27, 8, 40000)
            ids, rec = oracle.trace_records(g, rays)
            moving = rec[ids == 2]
            assert len(moving) > 100 and np.isfinite(moving[:, 7:9]).all()       # finite, varying uv on the moving sphere
            assert np.unique(moving[:, 7]).size > 50
            assert (ids == abi.ID_MEDIUM_BASE).sum() > 20                         # the smoke is hit
        what = "uv lab %s accel %d" % ("/".join(order), accel)
        for schedule in (abi.SCHEDULE_AUTO, abi.SCHEDULE_TILES, abi.SCHEDULE_REFILL):
            assert_bit_exact(sc.render(96, 54, 6, schedule=schedule), ref, what + " schedule %d" % schedule)
        assert_bit_exact(sc.render(96, 54, 6, pipeline=abi.PIPELINE_WAVEFRONT), ref, what + " staged")
        assert_same_hits(oracle, g, sc, rays, what)


@pytest.mark.parametrize("boundary", ["box", "moving"])
def test_medium_inside_other_boundaries(oracle, boundary):
    """ConstantMedium::hit (include/ptMedium.h:31-55) searches its boundary twice, whatever the boundary is: a box (six
    rectangle tests each time) and a moving sphere take the general path, a plain sphere the one that forms its
    quadratic once.  Frames and closest hits against the oracle, list and tree, every scheduler."""
    order = UV_ORDERS[1]
    ref = None
    for accel in (abi.ACCEL_LIST, abi.ACCEL_LBVH):
        g = _uv_lab(order, accel, boundary=boundary)
        sc = pt.Scene(g)
        if ref is None:
            ref = oracle.render(g, 96, 54, 6)
            rays = oracle.record_rays(g, 48, 27, 8, 40000)
            ids, _ = oracle.trace_records(g, rays)
            assert (ids == abi.ID_MEDIUM_BASE).sum() > 20                         # the smoke is hit
        what = "smoke inside a %s, accel %d" % (boundary, accel)
        for schedule in (abi.SCHEDULE_AUTO, abi.SCHEDULE_TILES, abi.SCHEDULE_REFILL):
            assert_bit_exact(sc.render(96, 54, 6, schedule=schedule), ref, what + " schedule %d" % schedule)
        assert_bit_exact(sc.render(96, 54, 6, pipeline=abi.PIPELINE_WAVEFRONT), ref, what + " staged")
        assert_same_hits(oracle, g, sc, rays, what)


def test_moving_sphere_uv_outside_asin_domain(oracle):
    """RTNW final with the image on its moving sphere (|p.y| > 1: asinf gives NaN, the texel index comes
    from the host's float-to-int conversion of NaN) and on both smoke volumes."""
    tex = np.random.default_rng(8).integers(0, 256, (32, 64, 3), dtype=np.uint8)
    ref = None
    for accel in (abi.ACCEL_LIST, abi.ACCEL_LBVH):
        g = OwnedDesc(pt.builtin_scene("final", 16 / 9, earth_rgb=tex))
        image = next(i for i, t in enumerate(g.lists["textures"]) if t.type == abi.TEX_IMAGE)
        for m in g.lists["materials"]:
            if m.type == abi.MAT_ISOTROPIC:
                m.texture = image
        moving = next(s for s in g.lists["shapes"] if s.type == abi.SHAPE_MOVING_SPHERE)
        g.lists["materials"][moving.material].texture = image
        g.scalars["accel"] = accel
        g.finalize()
        if ref is None:
            ref = oracle.render(g, 96, 54, 6)
        assert_bit_exact(pt.Scene(g).render(96, 54, 6), ref, "final, image on the moving sphere and the smoke, accel %d" % accel)


def test_two_million_triangles_at_full_scale(oracle):
    """BASELINE configs[4] at its stated size: 2 000 000 triangles in the Cornell enclosure, 3840x2160.
    The library's LBVH (30-bit Morton codes with index tie-break over 2 000 006 leaves) against the
    oracle's in-order walk of the flat list on sampled camera rays, the bounce rays leaving their hit
    points, and rays falling onto the field from above; a small frame end to end; determinism and
    8 shards summing to the frame at 3840x2160."""
    nx, ny = 3840, 2160
    d = pt.trimesh_scene(1000, nx / ny)
    d.c.accel = abi.ACCEL_LBVH
    assert d.c.n_world == 2_000_006
    sc = pt.Scene(d)
    assert len(sc.lbvh_nodes()) == d.c.n_world - 1
    rng = np.random.default_rng(2025)
    # camera rays through the part of the frame that shows the room
    px = (rng.integers(ny // 8, 7 * ny // 8, 900) * nx + rng.integers(nx // 4, 3 * nx // 4, 900)).astype(np.uint32)
    cam = oracle.camera_rays(d, nx, ny, px)
    ids, rec = oracle.trace_records(d, cam)
    hit = ids >= 0
    assert hit.mean() > 0.8
    # bounce rays from those hit points: half into the hemisphere about the normal, half at the lamp
    n_b = int(hit.sum())
    bounce = np.zeros(n_b, RAY_DTYPE)
    p, n = rec[hit, 1:4], rec[hit, 4:7]
    bounce["o"] = p
    v = rng.normal(size=(n_b, 3)).astype(np.float32)
    v *= np.sign((v * n).sum(1, keepdims=True)).astype(np.float32)
    lamp = np.stack([rng.uniform(213, 343, n_b), np.full(n_b, 554.0), rng.uniform(227, 332, n_b)], 1).astype(np.float32)
    to_lamp = rng.random(n_b) < 0.5
    bounce["d"] = np.where(to_lamp[:, None], lamp - p, v * rng.uniform(0.2, 30.0, (n_b, 1)).astype(np.float32))
    bounce["time"] = cam["time"][hit]
    bounce["seed"] = np.arange(n_b, dtype=np.uint32)
    # rays falling onto the field: the un-normalised triangle t competes with the floor and the walls
    n_f = 1000
    fall = np.zeros(n_f, RAY_DTYPE)
    fall["o"] = np.stack([rng.uniform(60, 495, n_f), rng.uniform(300, 550, n_f), rng.uniform(60, 495, n_f)], 1).astype(np.float32)
    dirs = rng.normal(size=(n_f, 3)).astype(np.float32)
    dirs[:, 1] = -np.abs(dirs[:, 1]) - 0.3
    fall["d"] = dirs * (10.0 ** rng.uniform(-1.5, 1.5, (n_f, 1))).astype(np.float32)
    fall["seed"] = np.arange(n_f, dtype=np.uint32)
    rays = np.concatenate([cam, bounce, fall])
    assert len(rays) >= 2000
    ids_o = assert_same_hits(oracle, d, sc, rays, "2M triangles")
    d.c.accel = abi.ACCEL_LBVH_WIDE                              # the same tree folded to four children per node
    wide = pt.Scene(d)
    ids_w, t_w = wide.trace_closest(rays)
    np.testing.assert_array_equal(ids_w, ids_o)
    assert_bit_exact(wide.render(480, 270, 1), sc.render(480, 270, 1), "2M triangles, 4-wide against binary")
    del wide
    d.c.accel = abi.ACCEL_LBVH
    tri = np.array([d.c.shapes[int(i)].type == abi.SHAPE_TRIANGLE for i in ids_o[ids_o >= 0]])
    assert tri.sum() > 250 and np.unique(ids_o[ids_o >= 0][tri]).size > 200
    # a small frame end to end (every ray of it walks the 2 000 006-entry list in the oracle)
    assert_bit_exact(sc.render(32, 18, 1), oracle.render(d, 32, 18, 1, threads=16), "2M triangles 32x18x1")
    # full-size properties
    a = sc.render(nx, ny, 1)
    assert sc.stats().schedule == abi.SCHEDULE_WALK
    assert_bit_exact(sc.render(nx, ny, 1), a, "determinism")
    assert_bit_exact(sc.render(nx, ny, 1, schedule=abi.SCHEDULE_TILES), a, "tiles schedule")
    acc = np.zeros_like(a)
    for s in range(8):
        acc += sc.render(nx, ny, 1, shard_index=s, shard_count=8)
    assert_bit_exact(acc, a, "8 shards")


def test_staged_pipeline_reports_what_its_stages_did(oracle):
    """pt_render_stage_stats after a render on the staged pipeline: the rays its extend stage searched and the hits
    its material queues shaded are the counting oracle's closest-hit searches and scatters for the same frame,
    queue occupancy and the stages' time shares are fractions, and the fused pipeline leaves the counters alone."""
    from tests import _oracle_stats
    d = pt.builtin_scene("cornell", 96 / 64)
    sc = pt.Scene(d)
    with pytest.raises(pt.PtError):
        sc.stage_stats()                                            # nothing has run on the staged pipeline yet
    img = sc.render(96, 64, 6, pipeline=abi.PIPELINE_WAVEFRONT)
    with pytest.raises(pt.PtError):
        sc.stage_stats()                                            # ... with the counters asked for
    assert np.array_equal(sc.render(96, 64, 6, pipeline=abi.PIPELINE_WAVEFRONT, stage_stats=1).view(np.uint32), img.view(np.uint32))
    st = sc.stage_stats()
    cnt, ref = _oracle_stats.counters(d, 96, 64, 6)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert st["groups"] == (96 // 8) * (64 // 8) * 64 // 256
    assert st["extend_lanes"] == cnt["world_hits"]
    assert st["shade_lanes"] == [cnt["diffuse_scatters"], 0, 0, 0]   # a Cornell box is Lambertian walls and a lamp (which ends a path in the extend stage)
    assert st["rounds"] >= st["groups"] and st["extend_waves"] >= st["rounds"]
    assert 0.3 < st["extend_occupancy"] <= 1.0 and 0.2 < st["shade_occupancy"] <= 1.0
    assert 0.05 < st["extend_time_share"] < 0.95
    sc.render(96, 64, 2)                                            # the fused pipeline, and a staged render that does not count:
    sc.render(96, 64, 2, pipeline=abi.PIPELINE_WAVEFRONT)           # the counters stay what they were
    assert sc.stage_stats() == st
    # every material queue of the sphere scene fills: Lambertian, metal, glass
    d2 = pt.builtin_scene("random", 2.0, accel=abi.ACCEL_LBVH)
    sc2 = pt.Scene(d2)
    sc2.render(160, 80, 4, pipeline=abi.PIPELINE_WAVEFRONT, stage_stats=1)
    st2 = sc2.stage_stats()
    cnt2, _ = _oracle_stats.counters(d2, 160, 80, 4)
    assert st2["extend_lanes"] == cnt2["world_hits"]
    assert all(n > 0 for n in st2["shade_lanes"][:3]) and st2["shade_lanes"][3] == 0
    assert sum(st2["shade_lanes"]) <= st2["extend_lanes"]
