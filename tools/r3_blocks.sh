#!/bin/bash
# All-features kernels over lists and the binary tree at 3 (tree build) / 4 / 5 workgroups per CU, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for lib in "" tools/exp/libptcore_a4.so tools/exp/libptcore_a5.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  q simple_light 1024 512 256
  q cornell_spheres 1024 1024 64
  q cornell 1024 1024 64 binary
done
