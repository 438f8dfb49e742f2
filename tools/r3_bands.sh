#!/bin/bash
# Pixel slots of the persistent-lane schedules from eight counters (blocks of N wave tiles) against one sweep, same box.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for b in -1 64 -1 64; do
  echo "== bands $b"
  q random 1920 1080 64 bands $b
  q random 1920 1080 8 bands $b
  q final 1920 1080 64 bands $b
  q final 1920 1080 8 bands $b
done
