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
EXPECTED = {"pt_k_cornell": 3, "pt_k_spheres": 2, "pt_wavefront": 13}     # the staged pipeline's chunk kernel: one per scene variant


@pytest.fixture(scope="module")
def assembled(tmp_path_factory):
    """The compiler's assembly of every kernel translation unit, made once for the tests below."""
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    out = tmp_path_factory.mktemp("kernel_asm")
    units = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(ROOT, "gpupathtracer_amd", "csrc", "pt_k_*.hip")))
    assert len(units) >= 11
    units.append("pt_wavefront")

    def assemble(unit):
        src = os.path.join(ROOT, "gpupathtracer_amd", "csrc", unit + ".hip")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-S", "--cuda-device-only", "-o", str(out / (unit + ".hip.s")), src], stderr=subprocess.DEVNULL)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(assemble, units))
    return out, units


def test_chunk_kernels_keep_ticket_and_barriers_in_one_loop(assembled):
    tmp_path, units = assembled
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_chunk_kernel_loops.py"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for unit in units:
        ok = [l for l in r.stdout.splitlines() if l.startswith("ok") and (" " + unit + ".hip.s ") in l.replace("  ", " ")]
        assert len(ok) == EXPECTED.get(unit, 1), "%s: %d\n%s" % (unit, len(ok), r.stdout)


def test_list_walks_read_the_scene_through_the_scalar_unit(assembled):
    """A list walk reads the same entry in every lane.  The persistent kernels (atomic loads and stores, s_sleep,
    s_setprio, clock reads) once lost every scalar load to the compiler's caution and walked the list through the vector
    unit, a third slower; the scene is read through the constant address space since.  Every kernel that walks a list
    without a tree must keep scalar loads inside its loops, whatever its schedule."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_scalar_scene_loads import loads_in_loops
    tmp_path, _ = assembled
    spheres = (tmp_path / "pt_k_spheres.hip.s").read_text()
    cornell = (tmp_path / "pt_k_cornell.hip.s").read_text()
    seen = 0
    for txt, needle in ((spheres, "k_render_tilesILj153603"), (spheres, "k_render_refillILj153603"), (spheres, "k_render_chunksILj153603"),
                        (spheres, "k_render_passILj153603"),
                        (cornell, "k_render_tilesILj556EN3ptd8AnyWorld"), (cornell, "k_render_refillILj556EN3ptd8AnyWorld"),
                        (cornell, "k_render_chunksILj556EN3ptd8AnyWorld")):
        found = loads_in_loops(txt, needle)
        assert len(found) == 1, needle
        name, scalar, vector = found[0]
        assert scalar >= 4, "%s: %d scalar loads in its loops (%d vector)" % (name, scalar, vector)
        seen += 1
    assert seen == 7


def test_the_loop_check_tells_a_stray_barrier_loop_from_block_placement(tmp_path):
    """The checker itself, on hand-written assembly: (a) the sound shape -- ticket and barriers in one loop, with a cold
    block laid out behind the loop that jumps back into its body (a backward branch that is no loop); (b) the shape
    that hangs -- an inner loop around the barriers that the ticket draw is not part of."""
    head = "_Z15k_render_chunksILj1EEvv:\n\ts_mov_b32 s0, 0\n"
    good = head + """.LBB0_1:
\tglobal_atomic_add v1, v0, v1, s[2:3] sc0
\ts_barrier
\ts_cbranch_scc1 .LBB0_9
.LBB0_2:
\tv_mov_b32 v0, v1
\ts_barrier
\ts_cbranch_scc0 .LBB0_1
\ts_branch .LBB0_10
.LBB0_9:
\tv_mov_b32 v2, v1
\ts_branch .LBB0_2
.LBB0_10:
\ts_endpgm
"""
    bad = head + """.LBB0_1:
\tglobal_atomic_add v1, v0, v1, s[2:3] sc0
.LBB0_2:
\ts_barrier
\tv_mov_b32 v0, v1
\ts_barrier
\ts_cbranch_scc0 .LBB0_2
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
"""
    for name, text, want in (("good", good, 0), ("bad", bad, 1)):
        d = tmp_path / name
        d.mkdir()
        (d / "k.s").write_text(text)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_chunk_kernel_loops.py"), str(d)], capture_output=True, text=True)
        assert r.returncode == want, r.stdout + r.stderr
        assert ("BAD" in r.stdout) == bool(want)
