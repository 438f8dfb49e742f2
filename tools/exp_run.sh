#!/bin/bash
set -e
Q="timeout -k 10 120 python tools/quick_time.py"
$Q random 1920 1080 64 lbvh
$Q final 1920 1080 64 lbvh
$Q random 1920 1080 64 lbvh schedule 1
$Q random 1920 1080 64 wide
timeout -k 10 600 python tests/make_work_model.py > gpurun_out/work_model.log 2>&1 && cp bench/work_model.json gpurun_out/work_model.json
