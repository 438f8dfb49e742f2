#!/bin/bash
# What AUTO picks now, timed.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q trimesh1000 3840 2160 64 lbvh
q trimesh1000 3840 2160 16 lbvh
q trimesh100 1920 1080 64 lbvh
q random 1920 1080 1024 lbvh
