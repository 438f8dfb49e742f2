#!/bin/bash
# Standard timing set (kernel ms, best of three): the four benchmark configurations at reduced sample counts.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q random 1920 1080 256 lbvh
q final 1920 1080 64 lbvh
q final 1920 1080 64
q trimesh1000 3840 2160 16 lbvh
q cornell 1024 1024 512
