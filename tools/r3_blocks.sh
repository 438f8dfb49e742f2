#!/bin/bash
# The final scene's chunk kernel over the 4-wide tree with 4 (tree build) / 2 / 1 waves per workgroup, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_cw2.so tools/exp/libptcore_cw1.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q final 1920 1080 256 lbvh
  q final 1920 1080 256 lbvh schedule 4 chunk 32
done
