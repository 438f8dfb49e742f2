#!/bin/bash
# The staged pipeline in chunks of samples, tree families.
set -e
T="timeout -k 10"
q() { $T 120 python3 tools/quick_time.py "$@"; }
q random 1920 1080 256 lbvh wavefront
q random 1920 1080 256 lbvh wavefront chunk 256
q random 1920 1080 1024 lbvh wavefront
q final 1920 1080 256 lbvh wavefront
q final 1920 1080 256 lbvh wavefront chunk 256
q final 1920 1080 256 wavefront
q final 1920 1080 256 wavefront chunk 256
q random 1920 1080 128 wavefront
q random 1920 1080 128 wavefront chunk 128
q trimesh1000 3840 2160 16 lbvh wavefront
q cornell 1024 1024 4096 wavefront
