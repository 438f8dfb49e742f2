#!/bin/bash
# Standard timing set, tree build against tools/exp/libptcore_prev.so, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_prev.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q cornell 1024 1024 4096
  q cornell 1024 1024 512 schedule 1
  q cornell 1024 1024 4096 shards 2
  q random 1920 1080 256 lbvh
  q random 1920 1080 64
  q final 1920 1080 256 lbvh
  q final 1920 1080 64
  q trimesh1000 3840 2160 16 lbvh
  q cornell 1024 1024 512 wavefront
  q random 1920 1080 64 lbvh wavefront
done
