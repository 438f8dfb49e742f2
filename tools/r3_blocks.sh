#!/bin/bash
# Staged kernels with the trimmed LDS records at 5 (tree build) and 6 workgroups per CU, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_w6.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q cornell 1024 1024 4096 wavefront
  q cornell 1024 1024 512 wavefront
  q cornell_spheres 1024 1024 256 wavefront
  q random 1920 1080 32 wavefront
done
