#!/bin/bash
# Collects a round's measurements on the GPU box into gpurun_out/<round>/ (copy what is to be kept
# into profiles/<round>/ afterwards):  tools/collect_profiles.sh r02
# Every GPU step runs under its own time limit; the script stops at the first failure.
set -e
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
T="timeout -k 10"
stats_csv() { find "$1" -name '*kernel_stats.csv' | head -1; }

echo "== counters of the headline kernel (bench.py quotes their HBM traffic while the stamp matches the library)"
$T 900 python3 tools/pmc_collect.py $O/pmc_cornell k_render_chunks -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-staged > $O/pmc_cornell.log
cp $O/pmc_cornell/summary.json $O/pmc_cornell1024x4096_k_render_chunks.json
mkdir -p profiles/$R && cp $O/pmc_cornell1024x4096_k_render_chunks.json profiles/$R/
echo "== counters"
$T 900 python3 tools/pmc_collect.py $O/pmc_shard k_render_tiles -- python3 tools/shard_render.py 8 512 > $O/pmc_shard.log
cp $O/pmc_shard/summary.json $O/pmc_cornell_eighth_shard.json
$T 900 python3 tools/pmc_collect.py $O/pmc_c1 k_render_walk -- python3 bench/configs.py --only c1 --quick > $O/pmc_c1.log
cp $O/pmc_c1/summary.json $O/pmc_c1_rtiow_lbvh_walk.json
$T 900 python3 tools/pmc_collect.py $O/pmc_c3lbvh k_render_tiles -- python3 bench/configs.py --only c3lbvh --quick > $O/pmc_c3lbvh.log
cp $O/pmc_c3lbvh/summary.json $O/pmc_c3_final_lbvh.json
$T 900 python3 tools/pmc_collect.py $O/pmc_c3 k_render_refill -- python3 bench/configs.py --only c3 --quick > $O/pmc_c3.log
cp $O/pmc_c3/summary.json $O/pmc_c3_final_list.json
$T 900 python3 tools/pmc_collect.py $O/pmc_c4 k_render_walk -- python3 bench/configs.py --only c4 --quick > $O/pmc_c4.log
cp $O/pmc_c4/summary.json $O/pmc_c4_trimesh_walk.json
cp $O/pmc_c4_trimesh_walk.json profiles/$R/          # bench/configs.py quotes its measured GB/s beside the algorithmic ones
echo "== bench.py default"
$T 600 python3 bench.py > $O/bench_cornell1024x4096_default.json
cat $O/bench_cornell1024x4096_default.json
echo "== bench.py under rocprofv3 --kernel-trace --stats"
$T 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline > $O/bench_cornell1024x4096_under_rocprof.json
cp "$(stats_csv $O/prof_bench)" $O/bench_cornell1024x4096_kernel_stats.csv
echo "== configs"
$T 600 python3 bench/configs.py > $O/configs_1gpu.jsonl
$T 600 rocprofv3 --kernel-trace --stats -d $O/prof_configs --output-format csv -- python3 bench/configs.py > $O/configs_1gpu_under_rocprof.jsonl
cp "$(stats_csv $O/prof_configs)" $O/configs_1gpu_kernel_stats.csv
echo "== sharding estimates"
$T 600 python3 tools/sharding_estimates.py > $O/sharding_estimates.json
echo "== staged pipeline"
$T 900 python3 tools/pmc_collect.py $O/pmc_wf k_wf_chunks -- python3 tools/wavefront_render.py 512 > $O/pmc_wf.log
cp $O/pmc_wf/summary.json $O/pmc_staged_cornell1024x512_k_wf_chunks.json
$T 900 python3 tools/pmc_collect.py $O/pmc_wf_c1 k_wf_groups -- python3 bench/configs.py --only c1 --quick --pipeline wavefront > $O/pmc_wf_c1.log
cp $O/pmc_wf_c1/summary.json $O/pmc_staged_c1_rtiow_lbvh_k_wf_groups.json
$T 600 python3 bench/configs.py --pipeline wavefront > $O/configs_1gpu_staged.jsonl
$T 300 rocprofv3 --kernel-trace --stats -d $O/prof_wf --output-format csv -- python3 tools/wavefront_render.py 4096 > $O/staged_cornell1024x4096.txt
cp "$(stats_csv $O/prof_wf)" $O/staged_cornell1024x4096_kernel_stats.csv
rm -rf $O/prof_bench $O/prof_configs $O/prof_wf $O/pmc_*/
echo "== done"
ls $O
