#!/bin/bash
# One fuzz world under several builds of the library.
T="timeout -k 10"
for lib in "" tools/exp/libptcore_prev.so tools/exp/libptcore_r02.so; do
  export PT_LIBPTCORE_PATH=$lib
  [ -z "$lib" ] && unset PT_LIBPTCORE_PATH
  echo "== lib: ${lib:-tree}"
  PT_FUZZ_FIRST=${SEED:-100401} PT_FUZZ_TEXTURED_WORLDS=1 $T 300 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread -k "random_textured" 2>&1 | grep -E "AssertionError: |passed|failed" | head -3
done
