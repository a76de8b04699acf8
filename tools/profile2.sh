#!/bin/bash
# extra PMC passes for the render kernel (memory pipeline view)
TAG=${1:-p2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/d -- $CMD > $OUT/d.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr --output-format csv -d $OUT/e -- $CMD > $OUT/e.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/f -- $CMD > $OUT/f.log 2>&1
for x in a b c d e f; do tail -2 $OUT/$x.log | cut -c1-200; done
