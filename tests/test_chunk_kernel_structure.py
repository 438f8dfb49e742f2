"""The chunked schedule's kernel must keep its loop shape through the compiler: its persistent loop pairs two
barriers per round with one ticket draw by thread 0, and a compiler that gives thread 0 a loop of its own (seen once)
leaves the other lanes of its wave circling through the barriers for ever -- on the GPU that is a hang, so the
check runs here, on the assembly of the headline's specialisations (no GPU needed, about a minute of hipcc)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


# the specialisations PT_SCHEDULE_AUTO runs in chunks: the rectangle scenes' three (generic walk, compiled-in walk, its
# branchless form) and the final scene's kinds through the 4-wide tree
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("unit,kernels", [("pt_k_cornell", 3), ("pt_k_final_lbvh4", 1)])
def test_chunk_kernels_keep_ticket_and_barriers_in_one_loop(tmp_path, unit, kernels):
    src = os.path.join(ROOT, "gpupathtracer_amd", "csrc", unit + ".hip")
    out = tmp_path / (unit + ".hip.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-S", "--cuda-device-only", "-o", str(out), src], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_chunk_kernel_loops.py"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("ok")]
    assert len(lines) == kernels, r.stdout
