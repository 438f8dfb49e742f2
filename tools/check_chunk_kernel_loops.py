#!/usr/bin/env python3
"""Structure check of every k_render_chunks specialisation in the compiler's assembly (no GPU needed).

The kernel's persistent loop pairs two barriers per round with one ticket draw by thread 0.  A compiler that rearranges
the loop for thread 0 alone (it did, for a first version of the source) leaves the other lanes circling through the
barriers on a stale ticket, for ever.  The check: every loop (backward branch) of the kernel that holds an s_barrier
also holds the ticket's global_atomic_add.

    mkdir -p /tmp/chk && cd gpupathtracer_amd/csrc && for f in pt_k_*.hip; do \
        hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -S --cuda-device-only -o /tmp/chk/$f.s $f; done
    python tools/check_chunk_kernel_loops.py /tmp/chk
"""
import re, sys, glob, subprocess
bad = 0
for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else '/tmp/chk') + '/*.s')):
    txt = open(f).read()
    # split into functions
    for m in re.finditer(r'^(_Z15k_render_chunks\S*):[^\n]*\n(.*?)\n\ts_endpgm', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.split('\n')
        # label positions
        pos = {}
        for i, l in enumerate(lines):
            mm = re.match(r'^(\.LBB\d+_\d+):', l)
            if mm: pos[mm.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            mm = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in pos and pos[mm.group(1)] <= i:
                loops.append((pos[mm.group(1)], i))
        atom = [i for i, l in enumerate(lines) if 'global_atomic_add' in l]
        bars = [i for i, l in enumerate(lines) if re.match(r'\s+s_barrier', l)]
        # the ticket atomics: possibly one in the peeled first round and one in the loop
        ok = True
        why = ''
        for lo, hi in loops:
            has_bar = any(lo <= b <= hi for b in bars)
            has_atom = any(lo <= a <= hi for a in atom)
            if has_bar and not has_atom:
                ok = False; why = 'loop %d-%d holds a barrier but no ticket' % (lo, hi)
        dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()
        print(('ok  ' if ok else 'BAD ') + f.split('/')[-1] + '  ' + dem[:90] + '  atomics %d barriers %d loops %d %s' % (len(atom), len(bars), len(loops), why))
        bad += (not ok)
print('bad:', bad)
sys.exit(1 if bad else 0)
