#!/usr/bin/env python3
"""Structure check of every k_render_chunks and k_wf_chunks specialisation in the compiler's assembly (no GPU needed).

The kernel's persistent loop pairs two barriers per round with one ticket draw by thread 0.  A compiler that rearranges
the loop for thread 0 alone (it did, for a first version of the source) leaves the other lanes circling through the
barriers on a stale ticket, for ever.  The check: every loop of the kernel that holds an s_barrier also holds the
ticket's global_atomic_add (the one that returns a value: the cost counters' atomics return nothing).  The staged
kernel (k_wf_chunks) runs rounds of stages inside a work item, barriers in a loop of their own: there the check is that
some loop holds both the ticket and a barrier.  For both kernels no loop may hold the ticket without a barrier.

Loops are the natural loops of the kernel's control-flow graph, built from the labels and branches of the assembly:
a branch to an earlier line is not enough to call something a loop (the compiler lays cold blocks out behind a loop
and jumps back into its body from them), so back edges are the edges whose target dominates their source.

    mkdir -p /tmp/chk && cd gpupathtracer_amd/csrc && for f in pt_k_*.hip; do \
        hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -S --cuda-device-only -o /tmp/chk/$f.s $f; done
    python tools/check_chunk_kernel_loops.py /tmp/chk
"""
import glob
import re
import subprocess
import sys


def blocks_of(lines):
    """Basic blocks: (first line, last line) ranges cut at labels and after branches, with successor lists."""
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            label_at[m.group(1)] = i
    starts = {0} | set(label_at.values())
    for i, l in enumerate(lines):
        if re.match(r'\s+s_c?branch', l) and i + 1 < len(lines):
            starts.add(i + 1)
    starts = sorted(starts)
    block_of_line = {}
    blocks = []
    for k, s in enumerate(starts):
        e = (starts[k + 1] if k + 1 < len(starts) else len(lines)) - 1
        blocks.append((s, e))
        for i in range(s, e + 1):
            block_of_line[i] = k
    succ = [[] for _ in blocks]
    for k, (s, e) in enumerate(blocks):
        last = None
        for i in range(e, s - 1, -1):                      # the last instruction of the block
            t = lines[i].strip()
            if t and not t.startswith(';') and not t.startswith('.') and not re.match(r'^\.?\w+:', t):
                last = t
                break
        falls = True
        if last:
            m = re.match(r's_(c?)branch\S*\s+(\.LBB\d+_\d+)', last)
            if m:
                if m.group(2) in label_at:
                    succ[k].append(block_of_line[label_at[m.group(2)]])
                falls = m.group(1) == 'c'
            elif last.startswith('s_endpgm'):
                falls = False
        if falls and k + 1 < len(blocks):
            succ[k].append(k + 1)
    return blocks, succ


def natural_loops(succ):
    n = len(succ)
    pred = [[] for _ in range(n)]
    for a in range(n):
        for b in succ[a]:
            pred[b].append(a)
    # reachable blocks in reverse post-order, then the iterative dominator sets
    order, seen = [], {0}
    stack = [(0, iter(succ[0]))]
    while stack:
        node, it = stack[-1]
        for b in it:
            if b not in seen:
                seen.add(b)
                stack.append((b, iter(succ[b])))
                break
        else:
            order.append(node)
            stack.pop()
    order.reverse()
    full = set(order)
    dom = {b: set(full) for b in order}
    dom[0] = {0}
    changed = True
    while changed:
        changed = False
        for b in order[1:]:
            ps = [dom[p] for p in pred[b] if p in full]
            new = (set.intersection(*ps) if ps else set()) | {b}
            if new != dom[b]:
                dom[b] = new
                changed = True
    loops = []
    for a in order:
        for h in succ[a]:
            if h in dom[a]:                                 # back edge a -> h
                body, work = {h, a}, [a]
                while work:
                    x = work.pop()
                    if x == h:
                        continue
                    for p in pred[x]:
                        if p in full and p not in body:
                            body.add(p)
                            work.append(p)
                loops.append((h, body))
    return loops


def main():
    bad = 0
    for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else '/tmp/chk') + '/*.s')):
        txt = open(f).read()
        for m in re.finditer(r'^(_Z\S*(?:15k_render_chunks|11k_wf_chunks)\S*):[^\n]*\n(.*?\n\ts_endpgm)', txt, re.S | re.M):
            name, body = m.group(1), m.group(2)
            staged = 'k_wf_chunks' in name          # its work item loops over rounds of stages: barriers in an inner loop of their own
            lines = body.split('\n')
            blocks, succ = blocks_of(lines)
            has_bar = [any(re.match(r'\s+s_barrier', lines[i]) for i in range(s, e + 1)) for s, e in blocks]
            has_ticket = [any('global_atomic_add' in lines[i] and ' sc0' in lines[i] for i in range(s, e + 1)) for s, e in blocks]
            loops = natural_loops(succ)
            ok, why = True, ''
            if sum(has_ticket) != 1:
                ok, why = False, 'expected one ticket draw, found %d' % sum(has_ticket)
            for h, members in loops:
                if not staged and any(has_bar[b] for b in members) and not any(has_ticket[b] for b in members):
                    ok = False
                    why = 'the loop headed at line %d holds a barrier but no ticket' % blocks[h][0]
                if any(has_ticket[b] for b in members) and not any(has_bar[b] for b in members):
                    ok = False
                    why = 'the loop headed at line %d holds the ticket but no barrier: thread 0 circles alone' % blocks[h][0]
            if staged and not any(any(has_ticket[b] for b in members) and any(has_bar[b] for b in members) for _, members in loops):
                ok, why = False, 'no loop holds both the ticket and a barrier'
            if not any(any(has_bar[b] for b in members) for _, members in loops):
                ok, why = False, 'no loop holds a barrier'
            dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()
            print(('ok  ' if ok else 'BAD ') + f.split('/')[-1] + '  ' + dem[:90] + '  barriers %d loops %d %s' % (sum(has_bar), len(loops), why))
            bad += (not ok)
    print('bad:', bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
