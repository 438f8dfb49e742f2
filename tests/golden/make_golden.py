"""Regenerates tests/golden/oracle_hashes.json: FNV-1a hashes of small renders made by the CPU
oracle in this image (NOT by the reference, which cannot be built here).  They guard the oracle
against accidental edits and let the GPU tests compare against a committed value."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpupathtracer_amd as pt   # noqa: E402
from tests import _oracle        # noqa: E402

CASES = [("cornell", 1.0, 64, 64, 8), ("random", 2.0, 80, 40, 4), ("simple_spheres", 2.0, 80, 40, 8),
         ("cornell_spheres", 1.0, 64, 64, 8), ("simple_light", 2.0, 80, 40, 8), ("final", 16 / 9, 64, 36, 8)]

if __name__ == "__main__":
    orc = _oracle.load()
    out = {"glibc": os.confstr("CS_GNU_LIBC_VERSION"), "images": {}}
    for name, aspect, nx, ny, ns in CASES:
        d = pt.builtin_scene(name, aspect)
        out["images"]["%s|%r|%d|%d|%d" % (name, aspect, nx, ny, ns)] = "%016x" % orc.fnv1a(orc.render(d, nx, ny, ns))
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_hashes.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
