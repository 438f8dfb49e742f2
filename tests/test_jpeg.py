"""pt_image_decode_jpeg, the decoder behind `--earthmap file.jpg` and image textures, against the reference's own
decoder (src/stb_image.h compiled as it lies into oracle/_ref/stb_decode) byte for byte: on the reference's
earthmap.jpg where the reference tree is present (its texels' hash is the committed pin,
tests/golden/earthmap_texels.json), and on files Pillow writes here -- every chroma layout, odd sizes, grey,
restart intervals, optimised Huffman tables.  No GPU involved."""
import hashlib
import io
import json
import os
import subprocess

import numpy as np
import pytest

import gpupathtracer_amd as pt
from gpupathtracer_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_DECODER = os.path.join(ROOT, "oracle", "_ref", "stb_decode")
EARTHMAP = os.path.join(os.environ.get("PT_REFERENCE_DIR", "/root/reference"), "earthmap.jpg")
PIN = json.load(open(os.path.join(HERE, "golden", "earthmap_texels.json")))


def reference_decode(path, tmp_path):
    raw = str(tmp_path / "ref.raw")
    nx, ny, nn = map(int, subprocess.run([REF_DECODER, path, raw], stdout=subprocess.PIPE, text=True, check=True).stdout.split())
    t = np.fromfile(raw, np.uint8).reshape(ny, nx, nn)
    return np.repeat(t, 3, axis=2) if nn == 1 else t


@pytest.mark.skipif(not os.path.exists(EARTHMAP), reason="the reference's earthmap.jpg is not on this machine")
def test_decoder_reproduces_the_reference_decoders_earth_map():
    data = open(EARTHMAP, "rb").read()
    assert hashlib.sha256(data).hexdigest() == PIN["file_sha256"]
    tex = api.decode_jpeg(data)
    assert tex.shape == (PIN["ny"], PIN["nx"], 3)
    assert [int(tex[..., k].sum()) for k in range(3)] == PIN["channel_sums"]
    assert hashlib.sha256(tex.tobytes()).hexdigest() == PIN["texels_sha256"]
    # the texels drive the final scene's earth sphere (src/ptMain.cu:396-402): a description built from them holds them unchanged
    d = pt.builtin_scene("final", 16 / 9, earth_rgb=tex)
    assert d.c.n_texels == tex.size and bytes(bytearray(d.c.texels[:12])) == tex.tobytes()[:12]


def _pictures():
    rng = np.random.default_rng(12)
    j, i = np.mgrid[0:99, 0:200]
    smooth = np.stack([(i * 255 // 199), (j * 255 // 98), ((i + j) * 255 // 297)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    edges = np.zeros((64, 48, 3), np.uint8); edges[::7] = 255; edges[:, ::5, 1] = 200; edges[20:40, 10:30] = (255, 0, 0)
    return {"smooth": smooth, "noise": noise, "edges": edges, "one": noise[:1, :1], "row": smooth[:1, :17], "column": smooth[:23, :1], "tile": noise[:16, :16]}


CASES = [(name, sub, q, kw) for name in ("smooth", "noise", "edges", "one", "row", "column", "tile") for sub, q, kw in
         ((0, 90, {}), (1, 75, {}), (2, 50, {}), (2, 95, {"optimize": True}), (0, 30, {"restart_marker_blocks": 3}))]


@pytest.mark.parametrize("name,subsampling,quality,extra", CASES)
def test_decoder_equals_the_reference_decoder_on_generated_files(tmp_path, name, subsampling, quality, extra):
    Image = pytest.importorskip("PIL.Image")
    path = str(tmp_path / "t.jpg")
    try:
        Image.fromarray(_pictures()[name]).save(path, "JPEG", quality=quality, subsampling=subsampling, **extra)
    except TypeError:
        pytest.skip("this Pillow does not write restart markers")
    data = open(path, "rb").read()
    mine = api.decode_jpeg(data)
    pil = np.asarray(Image.open(path).convert("RGB"))
    assert mine.shape == pil.shape
    # another decoder's inverse DCT and chroma filter round differently here and there, never by much
    # (with subsampled chroma the two enlarge it with different filters, which on noise is a large difference per texel)
    diff = np.abs(mine.astype(int) - pil.astype(int))
    assert diff.max() <= 4 if subsampling == 0 else diff.mean() <= 12
    if os.path.exists(REF_DECODER):
        assert np.array_equal(mine, reference_decode(path, tmp_path)), "%s subsampling %d quality %d %r" % (name, subsampling, quality, extra)


def test_grey_files_come_out_in_three_equal_channels(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    path = str(tmp_path / "g.jpg")
    Image.fromarray(_pictures()["smooth"][..., 0]).save(path, "JPEG", quality=85)
    mine = api.decode_jpeg(open(path, "rb").read())
    assert mine.shape == (99, 200, 3) and np.array_equal(mine[..., 0], mine[..., 1]) and np.array_equal(mine[..., 0], mine[..., 2])
    if os.path.exists(REF_DECODER):
        assert np.array_equal(mine, reference_decode(path, tmp_path))


def test_files_the_decoder_does_not_read_are_errors(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    path = str(tmp_path / "p.jpg")
    Image.fromarray(_pictures()["smooth"]).save(path, "JPEG", progressive=True)
    with pytest.raises(pt.PtError) as e:
        api.decode_jpeg(open(path, "rb").read())
    assert e.value.code == -1 and "progressive" in str(e.value)
    good = io.BytesIO()
    Image.fromarray(_pictures()["noise"]).save(good, "JPEG")
    for bad in (b"", b"\xff\xd8", b"not a jpeg at all", good.getvalue()[:200]):
        with pytest.raises(pt.PtError):
            api.decode_jpeg(bad)
    # damaged entropy data must not crash or read out of bounds: any outcome but a fault is acceptable
    blob = bytearray(good.getvalue())
    rng = np.random.default_rng(4)
    for _ in range(200):
        hurt = bytearray(blob)
        for k in rng.integers(20, len(hurt), 6):
            hurt[k] = int(rng.integers(0, 256))
        try:
            api.decode_jpeg(bytes(hurt))
        except pt.PtError:
            pass
