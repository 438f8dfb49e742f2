#!/bin/bash
# What AUTO picks now, timed.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q random 1920 1080 64
q random 1920 1080 256 lbvh
q random 1920 1080 1024 lbvh
q random 1920 1080 16 wavefront
q random 1920 1080 64 lbvh wavefront
q final 1920 1080 256 lbvh
q final 1920 1080 64
