"""The chunked schedule's kernel must keep its loop shape through the compiler: its persistent loop pairs two
barriers per round with one ticket draw by thread 0, and a compiler that gives thread 0 a loop of its own (seen once)
leaves the other lanes of its wave circling through the barriers for ever -- on the GPU that is a hang, so the
check runs here, on the assembly of every specialisation of the kernel (an explicit PT_SCHEDULE_CHUNKS reaches all
of them; no GPU needed, about half a minute of hipcc on eight cores)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
# k_render_chunks instantiations per translation unit: the rectangle scenes' three (generic walk, compiled-in walk,
# its branchless form), the sphere scenes' list walk and binary tree in one unit, one each elsewhere
EXPECTED = {"pt_k_cornell": 3, "pt_k_spheres": 2}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_chunk_kernels_keep_ticket_and_barriers_in_one_loop(tmp_path):
    units = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(ROOT, "gpupathtracer_amd", "csrc", "pt_k_*.hip")))
    assert len(units) >= 11

    def assemble(unit):
        src = os.path.join(ROOT, "gpupathtracer_amd", "csrc", unit + ".hip")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-S", "--cuda-device-only", "-o", str(tmp_path / (unit + ".hip.s")), src], stderr=subprocess.DEVNULL)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(assemble, units))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_chunk_kernel_loops.py"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for unit in units:
        ok = [l for l in r.stdout.splitlines() if l.startswith("ok") and (" " + unit + ".hip.s ") in l.replace("  ", " ")]
        assert len(ok) == EXPECTED.get(unit, 1), "%s: %d\n%s" % (unit, len(ok), r.stdout)
