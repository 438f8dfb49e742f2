#!/bin/bash
# Staged pipeline with a generate stage against the previous build, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
$T 900 python3 -u -m pytest tests -m gpu -x -q -k "wavefront or staged or stage or bench" --timeout 600 --timeout-method thread 2>&1 | tail -3
for lib in "" tools/exp/libptcore_prev.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q cornell 1024 1024 512 wavefront
  q random 1920 1080 64 lbvh wavefront
  q random 1920 1080 16 wavefront
  q final 1920 1080 16 lbvh wavefront
  q final 1920 1080 16 wavefront
  q trimesh1000 3840 2160 4 lbvh wavefront
done
