#!/bin/bash
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@" | cut -c1-70; }
for w in 16 24 32 36 44; do for v in 6 8 12; do
  export PT_EXP_WALK_WAIT=$w PT_EXP_LEAF_VOTE=$v
  echo "== wait $w vote $v"
  q trimesh1000 3840 2160 64 lbvh
  q trimesh1000 3840 2160 16 lbvh
done; done
export PT_EXP_WALK_WAIT=40 PT_EXP_LEAF_VOTE=1
echo "== wait 40, small meshes, vote 1"
q trimesh100 1920 1080 64 lbvh
q trimesh30 1920 1080 64 lbvh
export PT_EXP_WALK_WAIT=48
echo "== wait 48, small meshes, vote 1"
q trimesh100 1920 1080 64 lbvh
q trimesh30 1920 1080 64 lbvh
export PT_EXP_WALK_WAIT=32
echo "== wait 32, small meshes, vote 1"
q trimesh100 1920 1080 64 lbvh
q trimesh30 1920 1080 64 lbvh
