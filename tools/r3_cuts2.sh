#!/bin/bash
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for c in 16 8 4; do q trimesh1000 3840 2160 64 lbvh schedule 3 chunk $c; done
for c in 2 1; do q final 1920 1080 64 schedule 2 chunk $c; done
for c in 32 8 4 2; do q final 1920 1080 256 schedule 2 chunk $c; done
for c in 2 1; do q random 1920 1080 64 schedule 2 chunk $c; done
