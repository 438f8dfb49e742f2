#!/bin/bash
# Vector memory path counters of the triangle field's and the sphere scene's walk kernels.
set -e
T="timeout -k 10"
O=gpurun_out/r03
mkdir -p $O
export PMC_PASSES="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE;TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE;TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_REQUEST_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
$T 900 python3 tools/pmc_collect.py $O/tcp_c4 k_render_walk -- python3 bench/configs.py --only c4 --quick --no-cpu-baseline > $O/tcp_c4.log
cp $O/tcp_c4/summary.json $O/tcp_c4_trimesh_walk.json
$T 900 python3 tools/pmc_collect.py $O/tcp_c1 k_render_walk -- python3 bench/configs.py --only c1 --quick --no-cpu-baseline > $O/tcp_c1.log
cp $O/tcp_c1/summary.json $O/tcp_c1_rtiow_walk.json
