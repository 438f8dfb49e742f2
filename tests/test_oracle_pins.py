"""Pins of the CPU oracle.

The reference ships no tests, fixtures or golden vectors and cannot be built in this image
(it includes CUDA headers the image lacks), so the only recorded outputs of the reference's own
code are the known-answer values in the structural survey of the reference.  They are checked
here.  What they cover: both RNGs, the random_scene builder (draw order, object and material
counts), per-path work statistics of the Cornell and RTIOW integrators.  What they do NOT cover:
whole-image hashes -- the survey's FNV-1a image hashes could not be reproduced under any hash
convention tried, so image content beyond the statistics below is "parity unpinned".
"""
import json
import os

import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "survey_pins.json")))


def f32(v):
    return np.asarray(v, np.float32)


def test_pcg_stream_seed_12345(oracle):
    np.testing.assert_array_equal(oracle.pcg_stream(12345, 4), f32(PINS["pcg_seed_12345"]))


def test_pcg_stream_seed_0(oracle):
    # recorded by the survey in reverse order (unsequenced operator<< operands); same four values
    got = oracle.pcg_stream(0, 4)
    np.testing.assert_array_equal(np.sort(got), np.sort(f32(PINS["pcg_seed_0_any_order"])))
    np.testing.assert_array_equal(got[::-1], f32(PINS["pcg_seed_0_any_order"]))


def test_simple_rng_stream(oracle):
    np.testing.assert_array_equal(oracle.simple_rng_stream(42, 13, 4), f32(PINS["simple_rng_42_13"]))


def test_random_scene_matches_survey_counts_and_first_sphere():
    desc = pt.builtin_scene("random", 2.0)      # keep the owner alive while reading its arrays
    d = desc.c
    assert d.n_world == PINS["random_scene"]["n_objects"]
    kinds = [d.shapes[d.world[i].index].type for i in range(d.n_world)]
    mats = [d.materials[d.shapes[d.world[i].index].material].type for i in range(d.n_world)]
    assert kinds.count(abi.SHAPE_MOVING_SPHERE) == PINS["random_scene"]["moving_spheres"]
    assert mats.count(abi.MAT_METAL) == PINS["random_scene"]["metal"]
    assert mats.count(abi.MAT_DIELECTRIC) == PINS["random_scene"]["dielectric"]
    c = list(d.shapes[1].p)[:3]
    np.testing.assert_allclose(c, PINS["random_scene"]["first_small_sphere_centre"], rtol=0, atol=5e-6)


def test_cornell_scene_matches_survey_counts():
    desc = pt.builtin_scene("cornell", 1.0)
    d = desc.c
    assert d.n_world == 8
    rects = sum(6 if d.shapes[d.world[i].index].type == abi.SHAPE_BOX else 1 for i in range(d.n_world))
    assert rects == PINS["cornell"]["rectangles"]
    assert d.n_lights == 1 and d.shapes[d.lights[0]].type == abi.SHAPE_RECT_XZ


def test_final_scene_matches_survey_counts():
    desc = pt.builtin_scene("final", 16 / 9)
    d = desc.c
    assert d.n_world == PINS["final"]["top_level_objects_without_earth"]
    assert sum(1 for i in range(d.n_shapes) if d.shapes[i].type == abi.SHAPE_BOX) == PINS["final"]["bvh_boxes"]
    assert d.n_media == 2


@pytest.mark.parametrize("name,aspect,nx,ny,ns,key", [("cornell", 1.0, 128, 128, 32, "cornell_stats"),
                                                      ("random", 2.0, 200, 100, 16, "rtiow_stats")])
def test_work_statistics_agree_with_survey_probe(name, aspect, nx, ny, ns, key):
    """Closest-hit searches, RNG draws and diffuse scatters per path, counted by an instrumented
    build of the oracle, against the survey's gprof counts of the reference (3 significant digits;
    the survey ran Cornell at 256x256x32, statistics are resolution-independent to that precision)."""
    from tests import _oracle_stats
    st = _oracle_stats.run(name, aspect, nx, ny, ns)
    want = PINS[key]
    assert abs(st["world_hits_per_path"] - want["world_hits_per_path"]) < 0.02
    if "rng_draws_per_path" in want:
        assert abs(st["rng_draws_per_path"] - want["rng_draws_per_path"]) < 0.05
        assert abs(st["diffuse_scatters_per_path"] - want["diffuse_scatters_per_path"]) < 0.02


def test_work_model_file_matches_the_counting_oracle(oracle):
    """bench/work_model.json (the roofline's FLOP per path) is what the counting build measures, and
    the counting build renders the same image as the plain one.  The survey's probe counted 52.4
    rectangle tests per Cornell path."""
    from tests import _oracle_stats
    model = json.load(open(os.path.join(os.path.dirname(HERE), "bench", "work_model.json")))["configs"]["cornell"]
    d = pt.builtin_scene("cornell", 1.0)
    c, img = _oracle_stats.counters(d, 96, 96, 32)
    np.testing.assert_array_equal(img.view(np.uint32), oracle.render(d, 96, 96, 32).view(np.uint32))
    n = 96 * 96 * 32
    assert abs(c["flop"] / n - model["flop_per_path"]) < 0.02 * model["flop_per_path"]
    assert abs(c["rect_tests"] / n - 52.4) < 0.5
    assert abs(model["rect_tests_per_path"] - 52.4) < 0.2


def test_oracle_is_deterministic_across_thread_counts(oracle):
    d = pt.builtin_scene("cornell", 1.0)
    a = oracle.render(d, 48, 48, 4, threads=1)
    b = oracle.render(d, 48, 48, 4, threads=8)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_regression_hashes(oracle):
    """Guards the oracle against accidental edits: FNV-1a of small renders, recorded from this
    oracle itself (tests/golden/make_golden.py) with the image's glibc -- NOT reference output."""
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_hashes.json")))
    if gold["glibc"] != os.confstr("CS_GNU_LIBC_VERSION"):
        pytest.skip("hashes were recorded with another libm")
    for key, want in gold["images"].items():
        name, aspect, nx, ny, ns = key.split("|")
        d = pt.builtin_scene(name, float(aspect))
        img = oracle.render(d, int(nx), int(ny), int(ns))
        assert "%016x" % oracle.fnv1a(img) == want, key


@pytest.mark.xfail(reason="the survey's whole-image FNV-1a hashes have not been reproduced under any hash convention tried "
                          "(listed in golden/survey_pins.json); image content stays 'parity unpinned'", strict=False)
@pytest.mark.parametrize("key,name,aspect,nx,ny,ns", [("rtiow_200x100x16", "random", 2.0, 200, 100, 16),
                                                      ("cornell_128x128x16", "cornell", 1.0, 128, 128, 16),
                                                      ("final_192x108x16", "final", 16 / 9, 192, 108, 16)])
def test_survey_image_hashes(oracle, key, name, aspect, nx, ny, ns):
    """Not a gate: shows up as XPASS the day a convention (or a fix of the oracle) reproduces a hash."""
    img = oracle.render(pt.builtin_scene(name, aspect), nx, ny, ns)
    assert "%016x" % oracle.fnv1a(img) == PINS["image_hashes_not_reproduced"][key]


# ---- argument evaluation order of the reference's host compiler -------------------------------------------

def _eval_order():
    """Runs tests/evalorder/eval_order.cpp (shape-alikes of the reference's multi-draw expressions over
    a counting generator) under g++, the reference's host compiler, and returns {label: {slot: value}}."""
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(HERE, "evalorder", "eval_order.cpp")
    recorded = open(os.path.join(HERE, "evalorder", "eval_order_gxx.txt")).read()
    if shutil.which("g++"):
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "eval_order")
            subprocess.check_call(["g++", "-O2", "-std=c++11", src, "-o", exe])
            out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout
        assert out == recorded, "g++ here orders the draws differently from the recorded run"
    res = {}
    for line in recorded.splitlines():
        label, *tokens = line.split()
        vals, cur = {}, None
        for tok in tokens:                                    # "slot=value" opens a slot, a bare number adds to it
            if "=" in tok:
                cur, v = tok.split("=")
                vals[cur] = [float(v)]
            else:
                vals[cur].append(float(tok))
        res[label] = vals
    return res


def test_scene_builders_draw_in_the_order_gxx_evaluates(oracle):
    """random_scene (src/ptMain.cu:246-283) rebuilt here from the SimpleRng stream with the draw order
    the evaluation-order program reports for g++ -- z before x for the centre, the second centre before
    the colour, colour and albedo components z, y, x -- must equal the library's scene in every float."""
    order = _eval_order()
    assert order["center"] == {"x": [2.0], "z": [1.0]}
    assert order["moving_sphere"] == {"center1.y": [1.0], "colour": [67.0, 45.0, 23.0]}
    assert order["metal"] == {"albedo": [3.0, 2.0, 1.0]}
    assert order["cluster_sphere"] == {"centre": [3.0, 2.0, 1.0]}
    assert order["xz_rectangle_random"] == {"x": [2.0], "z": [1.0]}
    f = np.float32
    stream = iter(oracle.simple_rng_stream(42, 13, 20000))
    rand = lambda: f(next(stream))
    desc = pt.builtin_scene("random", 2.0)
    d = desc.c
    k = 1                                                         # world[0] is the ground
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = float(rand())
            rz = float(rand()); rx = float(rand())                # z is drawn first
            centre = (f(a + 0.9 * rx), f(0.2), f(b + 0.9 * rz))
            diff = (f(centre[0] - f(4)), f(centre[1] - f(0.2)), f(centre[2] - f(0)))
            length = np.sqrt(f(f(f(diff[0] * diff[0]) + f(diff[1] * diff[1])) + f(diff[2] * diff[2])))
            if not float(length) > 0.9:
                continue
            s = d.shapes[d.world[k].index]
            m = d.materials[s.material]
            k += 1
            assert (f(s.p[0]), f(s.p[1]), f(s.p[2])) == centre
            if choose < 0.8:
                c1y = f(centre[1] + f(0.5 * float(rand())))       # the second centre's draw comes before the colour's
                cz = f(rand() * rand()); cy = f(rand() * rand()); cx = f(rand() * rand())
                assert s.type == abi.SHAPE_MOVING_SPHERE and f(s.p[4]) == c1y and f(s.p[3]) == centre[0] and f(s.p[5]) == centre[2]
                t = d.textures[m.texture]
                assert m.type == abi.MAT_LAMBERTIAN and (f(t.color[0]), f(t.color[1]), f(t.color[2])) == (cx, cy, cz)
            elif choose < 0.95:
                mz = f(0.5 * float(rand())); my = f(0.5 * (1 + float(rand()))); mx = f(0.5 * (1 + float(rand())))
                assert s.type == abi.SHAPE_SPHERE and m.type == abi.MAT_METAL
                assert (f(m.albedo[0]), f(m.albedo[1]), f(m.albedo[2])) == (mx, my, mz)
            else:
                assert m.type == abi.MAT_DIELECTRIC
    assert k == d.n_world - 3                                     # the three large spheres close the list


def test_triangle_reports_the_unnormalised_t(oracle):
    """Known answer the survey of the reference recorded for Triangle::hit (src/ptTriangle.cu:64-72):
    a triangle whose determinant is 100, hit at true distance 5, reports rec.t = 500 while rec.p is
    the true point (it comes from the barycentrics)."""
    from gpupathtracer_amd import _abi as A
    from gpupathtracer_amd.api import RAY_DTYPE
    from tests._desc_tools import OwnedDesc
    g = OwnedDesc(pt.builtin_scene("trimesh", 1.0))
    tri = next(s for s in g.lists["shapes"] if s.type == abi.SHAPE_TRIANGLE)
    s = A.PtShape(type=abi.SHAPE_TRIANGLE, flags=0, material=tri.material, xform_first=0, xform_count=0, tri_uv=-1)
    for i, v in enumerate([0, 0, 0, 10, 0, 0, 0, 10, 0]):
        s.p[i] = float(v)
    g.lists["shapes"] = [s]
    g.lists["world"] = [A.PtObject(type=abi.OBJ_SHAPE, index=0)]
    g.lists["lights"] = []
    g.finalize()
    rays = np.zeros(1, RAY_DTYPE)
    rays["o"][0] = (1, 1, 5)
    rays["d"][0] = (0, 0, -1)
    ids, rec = oracle.trace_records(g, rays)
    assert ids[0] == 0 and rec[0, 0] == 500.0
    np.testing.assert_allclose(rec[0, 1:4], np.array([1, 1, 0], np.float32), rtol=0, atol=1e-6)      # 0.1f * 10 is one ulp above 1


def test_reference_bvh_of_the_final_scene_against_a_second_restatement(oracle):
    """The tree final() builds over its 400 boxes (BVH::BVH, src/ptBVH.cu:15-43; quickSort and its Lomuto
    partition with the last element as pivot and a strict '<' on bounds(0,0).min()[axis],
    src/ptQuickSort.cu:19-81; one SimpleRng draw per node for the axis), built a second time here,
    independently of the library's host code, and compared node for node: structure, leaf order, boxes."""
    f = np.float32
    desc = pt.builtin_scene("final", 16 / 9)
    d = desc.c
    boxes = [i for i in range(d.n_shapes) if d.shapes[i].type == abi.SHAPE_BOX]
    assert boxes == list(range(400))
    stream = iter(oracle.simple_rng_stream(42, 13, 400 + 2000))
    # the heights consumed one draw each (src/ptMain.cu:369-380): y1 = 100 * (rand + 0.01f)
    for i in boxes:
        y1 = f(f(100) * f(f(next(stream)) + f(0.01)))
        assert f(d.shapes[i].p[4]) == y1 and d.shapes[i].p[1] == 0.0
    lo = {i: np.array(d.shapes[i].p[0:3], f) for i in boxes}      # Box::bounds = (pmin, pmax), include/ptRectangle.h:219-223
    hi = {i: np.array(d.shapes[i].p[3:6], f) for i in boxes}

    def partition(lst, l, h, axis):
        x = lst[h]
        i = l - 1
        for j in range(l, h):
            if lo[lst[j]][axis] < lo[x][axis]:
                i += 1
                lst[i], lst[j] = lst[j], lst[i]
        lst[i + 1], lst[h] = lst[h], lst[i + 1]
        return i + 1

    def quick_sort(lst, axis):
        stack = [(0, len(lst) - 1)]
        while stack:
            l, h = stack.pop()
            p = partition(lst, l, h, axis)
            pushes = []
            if p - 1 > l:
                pushes.append((l, p - 1))
            if p + 1 < h:
                pushes.append((p + 1, h))
            stack.extend(pushes)                                  # the right part is pushed last and popped first, as there

    def build(lst):
        axis = int(f(3) * f(next(stream)))
        quick_sort(lst, axis)
        n = len(lst)
        if n == 1:
            left = right = ("leaf", lst[0])
        elif n == 2:
            left, right = ("leaf", lst[0]), ("leaf", lst[1])
        else:
            half = n // 2
            a, b = lst[:half], lst[half:]
            left = build(a)
            right = build(b)
            lst[:half], lst[half:] = a, b
        bl = lambda c: (lo[c[1]], hi[c[1]]) if c[0] == "leaf" else (c[3], c[4])
        (llo, lhi), (rlo, rhi) = bl(left), bl(right)
        return ("node", left, right, np.fmin(llo, rlo), np.fmax(lhi, rhi))

    mine = build(list(boxes))
    assert d.n_bvhs == 1

    def same(node, link):
        if node[0] == "leaf":
            assert link < 0 and ~link == node[1]
            return 1
        assert link >= 0
        nd = d.bvh_nodes[link]
        np.testing.assert_array_equal(np.array(nd.bmin[:], f).view(np.uint32), node[3].view(np.uint32))
        np.testing.assert_array_equal(np.array(nd.bmax[:], f).view(np.uint32), node[4].view(np.uint32))
        return same(node[1], nd.left) + same(node[2], nd.right)

    assert same(mine, d.bvhs[0].root) >= 400
