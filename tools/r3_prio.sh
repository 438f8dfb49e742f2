#!/bin/bash
# Measurement: CHUNKS with and without cost-based wave priority, against TILES, on the whole frame and on shards.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
for P in 0 1; do
  export PT_CHUNK_PRIORITY=$P
  echo "== PT_CHUNK_PRIORITY=$P"
  q cornell 1024 1024 4096 schedule 4
  q cornell 1024 1024 4096 schedule 4 shards 2
  q cornell 1024 1024 4096 schedule 4 shards 4
  q cornell 1024 1024 4096 schedule 4 shards 8
  q cornell 1024 1024 4096 schedule 4 chunk 128 shards 2
  q cornell 1024 1024 4096 schedule 4 chunk 128 shards 4
done
echo "== TILES"
q cornell 1024 1024 4096 schedule 1 shards 2
q cornell 1024 1024 4096 schedule 1 shards 4
q cornell 1024 1024 4096 schedule 1 shards 8
