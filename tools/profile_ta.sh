#!/bin/bash
# Run on the GPU box: what the CUs' vector-memory pipe (TA / TCP / TD) is doing during the render kernel -- busy and stall cycles, L1 hit rate,
# the L1's view of the L2 read latency -- next to the same counters on the TA micro-benchmark at known saturation (calibration).
# Counter passes only (no trace domains), at most four counters of one block per pass (more: "exceeds the capabilities of the hardware").
# Output: gpurun_out/prof_<tag>_ta/
TAG=${1:-r04}
CFG=${2:-headline}
OUT=gpurun_out/prof_${TAG}_ta
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-single --steps-per-launch 16 --config $CFG"
pass() {  # name, counters...: validated on the micro-benchmark first (a counter set the hardware refuses aborts in a second there, not after a bench run)
  local name=$1; shift
  rm -rf $OUT/$name $OUT/cal_$name
  if timeout -k 5 60 rocprofv3 --pmc "$@" --output-format csv -d $OUT/cal_$name -- tools/ubench/ta /dev/null 0.2 4:0:64 > $OUT/cal_$name.log 2>&1; then
    timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || echo "pass $name failed"
    echo "pass $name done"
  else
    echo "pass $name: counter set refused"
  fi
}
if [ "${3:-all}" = "all" ]; then
pass ta_busy TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum
pass tcp_tag TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum
fi
pass ta_stall TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp_lat TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp_gate TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
pass td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
pass sq_vmem SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES
python3 tools/summarize_ta.py $TAG
