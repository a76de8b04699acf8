#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into profiles/<tag>_*.
# Counters are collected in their own runs (never together with --sys-trace etc.), as the pool requires.
set -e
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
CFG=${2:-headline}
CMD="python3 bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-single --steps-per-launch 16 --config $CFG ${3:-}"   # every launch covers 16 frames
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
echo "kernel-trace done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1
echo "pmc sq1 done"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
echo "pmc sq2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc.log 2>&1
# the clock the kernel ran at (GRBM_GUI_ACTIVE / 8 / duration) and how long its CUs' vector-memory pipes had work (TA_TA_BUSY, summed over the 256 TAs), one pass
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d $OUT/pmc_grbm -- $CMD > $OUT/pmc_grbm.log 2>&1
echo "pmc mem done"
# traversal statistics of the same config (wave-steps = node fetches / 4, lanes and lines per step): the -DGLRTX_TRAV_STATS build, 8 frames in one launch
# (config 3 is a chain tree scanned as a list -- csrc/scan_asm.hip.h -- not traversed: there are no traversal steps to count)
if [ "$CFG" = "c3" ]; then
  echo "travstats skipped: config 3 is scanned as a list"
elif [ -f opengl-raytracer_amd/lib/libglrtx_stats.so ]; then
  GLRTX_TRAVSTATS_JSON=$OUT/travstats.json timeout -k 10 300 python3 tools/gpu_travstats.py $CFG 8 > $OUT/travstats.txt 2>&1 || echo "travstats failed"
fi
find $OUT -name "*.csv" | head -50
