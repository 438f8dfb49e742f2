#!/usr/bin/env python3
"""Scalar loads inside the loops of the render kernels, from the compiler's assembly (no GPU needed).

A list walk reads the same entry in every lane; its loads belong on the scalar unit.  Through an ordinary pointer the
compiler puts them there only while nothing in the kernel (an atomic load or store, s_sleep, s_setprio, a clock read)
stops it from showing that the memory is not written, and the persistent kernels have all of those: the scene is
therefore read through the constant address space (ScenePtr, pt_device.h).  This prints, per kernel named on the
command line, how many scalar and vector memory loads sit inside its loops, so that a change which sends the list walk
back to the vector unit (a third slower) shows up here rather than in a benchmark.

    python tools/check_scalar_scene_loads.py /tmp/chk/pt_k_spheres.hip.s k_render_refillILj153603 k_render_chunksILj153603
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_chunk_kernel_loops import blocks_of, natural_loops      # noqa: E402


def loads_in_loops(txt, kernel_substring):
    out = []
    for m in re.finditer(r'^(_Z\d+' + kernel_substring + r'\S*):[^\n]*\n(.*?\n\ts_endpgm)', txt, re.S | re.M):
        lines = m.group(2).split('\n')
        blocks, succ = blocks_of(lines)
        inside = set()
        for _, members in natural_loops(succ):
            inside |= members
        scalar = vector = 0
        for b in inside:
            for i in range(blocks[b][0], blocks[b][1] + 1):
                t = lines[i].strip()
                if t.startswith('s_load_') or t.startswith('s_buffer_load_'):
                    scalar += 1
                elif t.startswith('global_load_') or t.startswith('flat_load_') or t.startswith('buffer_load_'):
                    vector += 1
        out.append((m.group(1), scalar, vector))
    return out


def main():
    txt = open(sys.argv[1]).read()
    for k in sys.argv[2:]:
        for name, scalar, vector in loads_in_loops(txt, k):
            print("%s  scalar %d  vector %d" % (name[:70], scalar, vector))


if __name__ == "__main__":
    main()
