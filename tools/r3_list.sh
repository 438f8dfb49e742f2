#!/bin/bash
# REFILL cut and uncut with the scene read through the constant address space.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q cornell 1024 1024 4096
q random 1920 1080 64
q random 1920 1080 64 schedule 2 chunk 9999
q random 1920 1080 64 schedule 2 chunk 32
q random 1920 1080 64 schedule 2 chunk 16
q random 1920 1080 64 schedule 2 chunk 4
q final 1920 1080 64
q final 1920 1080 64 schedule 2 chunk 9999
q final 1920 1080 64 schedule 2 chunk 16
q final 1920 1080 64 schedule 2 chunk 4
q final 1920 1080 64 lbvh
q final 1920 1080 64 lbvh schedule 4 chunk 16
q final 1920 1080 64 lbvh schedule 3
