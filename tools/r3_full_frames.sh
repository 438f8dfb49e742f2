#!/bin/bash
# The benchmark configurations' frames against the oracle at their stated sizes (reduced samples where the oracle's rate asks for it).
set -e
O=gpurun_out/r03
mkdir -p $O
T="timeout -k 10"
F="python3 -u tests/full_frame_check.py"
$T 400 $F --scene cornell --size 1024 --spp 4096 --out $O/full_frame_parity_cornell1024x4096.json > $O/ffp_cornell.log 2>&1; tail -1 $O/ffp_cornell.log
$T 300 $F --scene final --earthmap --size 1920 --ny 1080 --spp 256 --lbvh --out $O/full_frame_parity_final_earthmap1920x1080x256_lbvh.json > $O/ffp_final_lbvh.log 2>&1; tail -1 $O/ffp_final_lbvh.log
$T 300 $F --scene final --earthmap --size 1920 --ny 1080 --spp 64 --out $O/full_frame_parity_final_earthmap1920x1080x64_list.json > $O/ffp_final_list.log 2>&1; tail -1 $O/ffp_final_list.log
$T 300 $F --scene random --size 1920 --ny 1080 --spp 32 --lbvh --out $O/full_frame_parity_rtiow1920x1080x32_lbvh.json > $O/ffp_rtiow_lbvh.log 2>&1; tail -1 $O/ffp_rtiow_lbvh.log
$T 300 $F --scene random --size 1920 --ny 1080 --spp 16 --out $O/full_frame_parity_rtiow1920x1080x16_list.json > $O/ffp_rtiow_list.log 2>&1; tail -1 $O/ffp_rtiow_list.log
$T 300 $F --trimesh 100 --size 480 --ny 270 --spp 16 --lbvh --out $O/full_frame_parity_trimesh20000_480x270x16_lbvh.json > $O/ffp_tri.log 2>&1; tail -1 $O/ffp_tri.log
$T 300 $F --scene cornell --size 1024 --spp 256 --wavefront --out $O/full_frame_parity_staged_cornell1024x256.json > $O/ffp_staged_cornell.log 2>&1; tail -1 $O/ffp_staged_cornell.log
$T 300 $F --scene final --earthmap --size 1920 --ny 1080 --spp 64 --lbvh --wavefront --out $O/full_frame_parity_staged_final_earthmap1920x1080x64_lbvh.json > $O/ffp_staged_final.log 2>&1; tail -1 $O/ffp_staged_final.log
$T 300 $F --scene random --size 1920 --ny 1080 --spp 16 --lbvh --wavefront --out $O/full_frame_parity_staged_rtiow1920x1080x16_lbvh.json > $O/ffp_staged_rtiow.log 2>&1; tail -1 $O/ffp_staged_rtiow.log
