#!/bin/bash
# The staged pipeline in chunks of samples on the headline frame.
set -e
T="timeout -k 10"
q() { $T 120 python3 tools/quick_time.py "$@"; }
q cornell 1024 1024 4096 wavefront
q cornell 1024 1024 4096 wavefront chunk 4096
q cornell 1024 1024 4096 wavefront chunk 1024
q cornell 1024 1024 4096 wavefront chunk 256
q cornell 1024 1024 4096 wavefront chunk 128
q cornell 1024 1024 4096 wavefront chunk 64
q cornell 1024 1024 512 wavefront
q cornell 1024 1024 512 wavefront chunk 512
q random 1920 1080 64 wavefront
q random 1920 1080 64 wavefront chunk 64
