#!/bin/bash
# Run on the GPU box: VALU issue-rate micro-benchmark + counter calibration of the lane-utilisation formula.
set -e
OUT=gpurun_out/ubench_valu
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 tools/ubench/valu $OUT/valu.jsonl > $OUT/valu.txt 2>&1
echo "ubench done"
for w in fma fma_half; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_$w -- tools/ubench/valu --only $w > $OUT/pmc_$w.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc2_$w -- tools/ubench/valu --only $w > $OUT/pmc2_$w.log 2>&1 || true
done
echo "pmc done"
find $OUT -name "*counter_collection.csv" | head
