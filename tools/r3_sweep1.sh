#!/bin/bash
# Measurement: the persistent-lane schedules with and without chunks of samples.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q random 1920 1080 256 lbvh schedule 3 chunk 100000
q random 1920 1080 256 lbvh schedule 3 chunk 64
q random 1920 1080 256 lbvh schedule 3 chunk 32
q random 1920 1080 256 lbvh schedule 3 chunk 16
q random 1920 1080 256 lbvh schedule 3 chunk 8
q random 1920 1080 256 lbvh schedule 3
q random 1920 1080 1024 lbvh schedule 3 chunk 100000
q random 1920 1080 1024 lbvh schedule 3
q random 1920 1080 1024 lbvh schedule 3 chunk 32
q final 1920 1080 64 schedule 2 chunk 100000
q final 1920 1080 64 schedule 2
q final 1920 1080 64 schedule 2 chunk 8
q final 1920 1080 256 schedule 2 chunk 100000
q final 1920 1080 256 schedule 2
q trimesh1000 3840 2160 16 lbvh schedule 3 chunk 100000
q trimesh1000 3840 2160 16 lbvh schedule 3 chunk 4
q trimesh1000 3840 2160 64 lbvh schedule 3 chunk 100000
q trimesh1000 3840 2160 64 lbvh schedule 3
q trimesh1000 3840 2160 64 lbvh schedule 3 chunk 8
q random 1920 1080 16 schedule 2 chunk 100000
q random 1920 1080 16 schedule 2 chunk 4
