#!/usr/bin/env python3
"""Writes tests/golden/earthmap_texels.json: size and SHA-256 of the texels the reference's own decoder
(src/stb_image.h, compiled as it lies by `make -C oracle ref` into oracle/_ref/stb_decode) produces for the
reference's earthmap.jpg, the image texture of its final scene (src/ptMain.cu:396-402).  The image itself is not
stored; the hash is what pt_image_decode_jpeg is held to.  Run in the build container, where /root/reference is."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PT_REFERENCE_DIR", "/root/reference")


def main():
    decoder = os.path.join(ROOT, "oracle", "_ref", "stb_decode")
    src = os.path.join(REF, "earthmap.jpg")
    if not (os.path.exists(decoder) and os.path.exists(src)):
        sys.exit("needs %s (make -C oracle ref) and %s" % (decoder, src))
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "earth.raw")
        nx, ny, nn = map(int, subprocess.run([decoder, src, raw], stdout=subprocess.PIPE, text=True, check=True).stdout.split())
        texels = open(raw, "rb").read()
    pin = {"file": "earthmap.jpg (reference asset; not stored here)", "file_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
           "nx": nx, "ny": ny, "channels": nn, "texels_sha256": hashlib.sha256(texels).hexdigest(),
           "channel_sums": [sum(texels[k::nn]) for k in range(nn)],
           "made_by": "oracle/_ref/stb_decode = the reference's src/stb_image.h (stb_image v2.15) compiled with gcc -O2, stbi_load(path, &nx, &ny, &n, 0)"}
    with open(os.path.join(HERE, "earthmap_texels.json"), "w") as f:
        json.dump(pin, f, indent=1)
        f.write("\n")
    print(pin)


if __name__ == "__main__":
    main()
