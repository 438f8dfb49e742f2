"""The C-ABI library loads on a CPU-only box and exports every symbol include/ptcore.h declares;
argument validation works without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = pt.library()
    hdr = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTS)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    assert C.sizeof(abi.PtShape) == 64
    assert C.sizeof(abi.PtRay) == 32
    assert C.sizeof(abi.PtBvhNode) == 32
    assert C.sizeof(abi.PtXform) == 16
    assert C.sizeof(abi.PtMaterial) == 24
    assert C.sizeof(abi.PtTexture) == 40
    assert C.sizeof(abi.PtRenderParams) == 64


def test_unknown_scene_is_an_error_with_message():
    with pytest.raises(pt.PtError) as e:
        pt.builtin_scene("no_such_scene", 1.0)
    assert e.value.code == abi.ERR_INVALID
    assert "no_such_scene" in str(e.value)


def test_scene_create_validates_before_touching_the_gpu():
    d = pt.builtin_scene("cornell", 1.0)
    keep = d.c.world[0].index
    d.c.world[0].index = 10_000          # out of range
    with pytest.raises(pt.PtError) as e:
        pt.Scene(d)
    assert e.value.code == abi.ERR_INVALID and "world" in str(e.value)
    d.c.world[0].index = keep
    m = d.c.shapes[0].material
    d.c.shapes[0].material = 999
    with pytest.raises(pt.PtError) as e:
        pt.Scene(d)
    assert e.value.code == abi.ERR_INVALID
    d.c.shapes[0].material = m


@pytest.mark.skipif(pt.device_count() > 0, reason="only meaningful without a GPU")
def test_no_device_is_reported_not_faked():
    d = pt.builtin_scene("cornell", 1.0)
    with pytest.raises(pt.PtError) as e:
        pt.Scene(d)
    assert e.value.code == abi.ERR_NO_DEVICE


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import gpupathtracer_amd.api as api
    monkeypatch.setattr(api, "_LIB", None)
    monkeypatch.setattr(api, "library_path", lambda: str(tmp_path / "libptcore.so"))
    with pytest.raises(ImportError):
        api.library()


def test_libm_host_build_matches_glibc(oracle):
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-130, 130, 400_000), rng.uniform(-7, 7, 400_000),
                        (10.0 ** rng.uniform(-38, 38, 200_000)) * rng.choice([-1, 1], 200_000),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 120.0, 0.78539816])]).astype(np.float32)
    for fn, arg in ((0, x), (1, x), (2, np.abs(x)), (3, np.clip(x, -4, 4))):
        a = pt.libm_eval(fn, arg, on_device=False)
        b = oracle.libm(fn, arg)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (fn, arg[~same][:5])
