#!/bin/bash
# Staged tree kernels in chunks: groups of 512 lanes (the library's choice for sphere and box trees) against 256, at 4 workgroups per CU.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
export PT_LIBPTCORE_PATH=tools/exp/libptcore_w4.so
for g in 512 256; do
  export PT_WF_GROUP=$g
  echo "== groups of $g lanes"
  q random 1920 1080 256 lbvh wavefront
  q final 1920 1080 256 lbvh wavefront
done
