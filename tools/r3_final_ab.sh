#!/bin/bash
# Final-scene timings, tree build against tools/exp/libptcore_prev.so, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_prev.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q final 1920 1080 256 lbvh
  q final 1920 1080 64
  q final 1920 1080 64 lbvh wavefront
done
