#!/bin/bash
# Tile-major chunk state against the pixel-major one (tools/exp/libptcore_prev.so), same box.
set -e
T="timeout -k 10"
mkdir -p gpurun_out/r03
$T 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "chunk" --timeout 300 --timeout-method thread 2>&1 | tail -3
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_prev.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q cornell 1024 1024 4096
  q cornell 1024 1024 4096 shards 2
  q cornell 1024 1024 4096 shards 4
  q cornell 1024 1024 512 shards 2 schedule 4
  q cornell 1024 1024 512 shards 2 schedule 1
  q final 1920 1080 64 lbvh
done
