#!/bin/bash
# tools/gpu_regime_counters.py under three counter passes; output gpurun_out/regime_pmc/
OUT=gpurun_out/regime_pmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 100 python3 tools/gpu_regime_counters.py 6 > $OUT/plain.txt 2>&1
timeout -k 5 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/utcl -- python3 tools/gpu_regime_counters.py 6 > $OUT/utcl.txt 2>&1 || echo "utcl pass failed"
timeout -k 5 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/lat -- python3 tools/gpu_regime_counters.py 6 > $OUT/lat.txt 2>&1 || echo "lat pass failed"
timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- python3 tools/gpu_regime_counters.py 6 > $OUT/tcc.txt 2>&1 || echo "tcc pass failed"
tail -3 $OUT/plain.txt $OUT/utcl.txt
find $OUT -name "*counter_collection.csv" | head
