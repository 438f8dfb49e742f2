#!/bin/bash
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q final 1920 1080 256 schedule 2 chunk 1
q final 1920 1080 256 lbvh schedule 2 chunk 1
q final 1920 1080 256 lbvh schedule 2 chunk 4
q final 1920 1080 256 lbvh
q cornell 1024 1024 512 schedule 2 chunk 1
q cornell 1024 1024 512 schedule 2 chunk 8
q cornell 1024 1024 512
