#!/bin/bash
# The chunk kernel's loop body against the tile kernel's, on a frame large enough that the schedule does not matter.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_prev.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q cornell 4096 4096 64 schedule 1
  q cornell 4096 4096 64 schedule 4 chunk 64
  q cornell 4096 4096 64 schedule 4 chunk 32
done
