#!/bin/bash
# The staged pipeline in chunks of samples on the headline frame.
set -e
T="timeout -k 10"
q() { $T 120 python3 tools/quick_time.py "$@"; }
q cornell 1024 1024 4096 wavefront
q cornell 1024 1024 4096 wavefront chunk 256
q cornell 1024 1024 4096 wavefront chunk 342
q cornell 1024 1024 512 wavefront
q cornell 1024 1024 4096 wavefront shards 2
q cornell 1024 1024 4096 wavefront shards 4
