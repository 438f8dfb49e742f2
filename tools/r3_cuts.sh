#!/bin/bash
# Chunk counts of the persistent-lane schedules re-measured with the slot counters per XCD.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for c in 256 64 32 16 8; do q random 1920 1080 256 lbvh schedule 3 chunk $c; done
for c in 1024 256 128 64 32; do q random 1920 1080 1024 lbvh schedule 3 chunk $c; done
for c in 64 32 16; do q trimesh1000 3840 2160 64 lbvh schedule 3 chunk $c; done
for c in 64 16 8 4 2; do q final 1920 1080 64 schedule 2 chunk $c; done
for c in 64 16 8 4 2; do q random 1920 1080 64 schedule 2 chunk $c; done
