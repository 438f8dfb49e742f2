#!/bin/bash
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for spp in 512 1024; do
  for sh in 2 3 4 6; do
    q cornell 1024 1024 $spp schedule 4 shards $sh
    q cornell 1024 1024 $spp schedule 1 shards $sh
  done
done
q final 1920 1080 256 lbvh schedule 4
q final 1920 1080 256 lbvh schedule 1
q final 1920 1080 256 lbvh schedule 4 shards 2
q final 1920 1080 256 lbvh schedule 1 shards 2
q final 1920 1080 256 lbvh schedule 4 shards 4
q final 1920 1080 256 lbvh schedule 1 shards 4
PT_CHUNK_PRIORITY=0 q final 1920 1080 256 lbvh schedule 4
