#!/bin/bash
# Run on the GPU box (through gpurun): vector-lane utilisation of the render kernel BY PHASE.
#   pass 1: rocprofv3 --pmc over tools/gpu_replay.py (libglrtx_raylog.so, `make diag`): the render kernel and, separately, the traverse phase replayed alone over
#           the recorded ray queues (pt_replay_traverse) -- SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 of each;
#   pass 2: the traversal-statistics build (libglrtx_stats.so): lanes per wave-step on the fork arm and on the leaf arm.
# tools/summarize_phase_util.py turns both into profiles/<tag>_lane_util.txt.  Counters are collected in their own run (no trace domains), as the pool requires.
set -e
TAG=${1:-r05}
CFG=${2:-headline}
OUT=gpurun_out/prof_${TAG}_phase
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp GLRTX_REPLAY_ORDERS=0
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 tools/gpu_replay.py $CFG 8 3 > $OUT/replay.log 2>&1
echo "pmc pass done"; tail -3 $OUT/replay.log
GLRTX_TRAVSTATS_JSON=$OUT/travstats.json timeout -k 10 300 python3 tools/gpu_travstats.py $CFG 8 > $OUT/travstats.txt 2>&1 || echo "travstats failed"
tail -5 $OUT/travstats.txt
