#!/bin/bash
# What AUTO picks now, timed.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q random 1920 1080 64
q random 1920 1080 16 wavefront
q cornell_spheres 1024 1024 64
q simple_spheres 1024 512 256
q simple_light 1024 512 256
q cornell 1024 1024 512 schedule 1
q cornell 1024 1024 4096
q trimesh30 480 270 16
