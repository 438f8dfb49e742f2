#!/bin/bash
# Builds tools/exp/libptcore_<tag>.so: the library with the named kernel translation units recompiled
# under extra compiler flags (kernel experiments; the result is selected with PT_LIBPTCORE_PATH).
#   tools/build_exp.sh <tag> "<extra flags>" <unit> [<unit> ...]      e.g.  tools/build_exp.sh c7 -DPT_EXP_BLOCKS=7 pt_k_cornell
set -e
cd "$(dirname "$0")/../gpupathtracer_amd/csrc"
tag=$1; flags=$2; shift 2
mkdir -p ../../tools/exp /tmp/ptexp_$tag
objs=""
for o in build/*.o; do
  keep=1
  for u in "$@"; do [ "$o" = "build/$u.o" ] && keep=0; done
  [ $keep = 1 ] && objs="$objs $o"
done
for u in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $flags -c -o /tmp/ptexp_$tag/$u.o $u.hip &
done
wait
new=""
for u in "$@"; do new="$new /tmp/ptexp_$tag/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/exp/libptcore_$tag.so $objs $new -L/opt/rocm/lib -lrccl
echo "built tools/exp/libptcore_$tag.so"
