#!/bin/bash
# Staged kernels at their table's workgroups per CU (tree build) / 5 / 6 everywhere, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_w5.so tools/exp/libptcore_w6.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q random 1920 1080 64 lbvh wavefront
  q final 1920 1080 64 lbvh wavefront
  q final 1920 1080 64 wavefront
  q random 1920 1080 16 wavefront
  q cornell 1024 1024 512 wavefront
  q trimesh1000 3840 2160 16 lbvh wavefront
done
echo "== fused, final and all-features kernels at their new caps"
q final 1920 1080 256
q final 1920 1080 64 binary
q cornell 1024 1024 64 lbvh
