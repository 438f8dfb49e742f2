#!/bin/bash
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q final 1920 1080 256 lbvh schedule 1
q final 1920 1080 256 lbvh schedule 4 chunk 32
q final 1920 1080 256 lbvh schedule 4 chunk 16
q final 1920 1080 256 lbvh schedule 4 chunk 8
