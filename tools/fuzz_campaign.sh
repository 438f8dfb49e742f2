#!/bin/bash
# Longer parity campaigns than the collected tests run: random rooms of the Cornell box's kinds, sphere worlds, triangle fields and random worlds of all kinds (plain, textured, at other sizes) against the
# oracle.  Each step under its own time limit; anything but a plain test failure ends the script.
# STEPS="rooms spheres" runs only the named campaigns (a GPU call is limited to twenty minutes).
want() { [ -z "$STEPS" ] || [[ " $STEPS " == *" $1 "* ]]; }
run() { local log=$1; shift; "$@" > "$log" 2>&1; local rc=$?; tail -2 "$log"; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: rc $rc"; exit $rc; fi; return 0; }
PY="python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread --maxfail 6"
mkdir -p gpurun_out
want rooms && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_ROOMS=${ROOMS:-20000} run gpurun_out/fuzz_rooms.log timeout -k 10 500 $PY -k "random_rooms"
want spheres && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_SPHERE_WORLDS=${SPHERES:-4000} run gpurun_out/fuzz_spheres.log timeout -k 10 500 $PY -k "random_sphere_worlds"
want triangles && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_TRIANGLE_WORLDS=${TRIANGLES:-3000} run gpurun_out/fuzz_triangles.log timeout -k 10 500 $PY -k "random_triangle_worlds"
want finals && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_FINAL_WORLDS=${FINALS:-3000} run gpurun_out/fuzz_final_kinds.log timeout -k 10 500 $PY -k "final_scenes_kinds"
want scaled && PT_FUZZ_SCALES=${SCALES:--38,-35,-33,-30,-27,-24,-18,-12,-6,6,12,18,24,30,36,40,44,48,52,56,60} PT_FUZZ_SCALED_WORLDS=${SCALED:-40} run gpurun_out/fuzz_scaled.log timeout -k 10 500 $PY -k "other_sizes"
want textured && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_TEXTURED_WORLDS=${TEXTURED:-4000} run gpurun_out/fuzz_textured.log timeout -k 10 500 $PY -k "random_textured"
want plain && PT_FUZZ_FIRST=${FIRST:-0} PT_FUZZ_WORLDS=${PLAIN:-6000} run gpurun_out/fuzz_plain.log timeout -k 10 500 $PY -k "random_worlds_match"
