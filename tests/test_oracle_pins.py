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
