#!/bin/bash
# Tree kernels at 6 (tree build) / 5 / 4 workgroups per CU, same box: tools/exp/libptcore_b5.so, _b4.so.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_b5.so tools/exp/libptcore_b4.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q random 1920 1080 256 lbvh
  q final 1920 1080 256 lbvh
  q final 1920 1080 64 lbvh schedule 1
  q trimesh1000 3840 2160 16 lbvh
done
