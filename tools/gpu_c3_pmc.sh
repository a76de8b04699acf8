#!/bin/bash
# Config 3 (the list scan) under two builds of the device library: instruction and wave counters of pt_render_wgwf, per frame.
# usage (GPU box): bash tools/gpu_c3_pmc.sh name=path/to/lib.so ...     -> gpurun_out/c3_pmc/<name>_{a,b}/, a table on stdout
export TMPDIR=/tmp
OUT=gpurun_out/c3_pmc; rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  n=${v%%=*}; lib=${v#*=}
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${n}_a -- python3 tools/gpu_abx.py --child $PWD/$lib c3 8 1 > $OUT/${n}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY --output-format csv -d $OUT/${n}_b -- python3 tools/gpu_abx.py --child $PWD/$lib c3 8 1 > $OUT/${n}_b.log 2>&1
  echo "$n done"
done
python3 - "$@" <<'P'
import csv, glob, sys, collections
rows = {}
for v in sys.argv[1:]:
    n = v.split("=")[0]
    acc = collections.defaultdict(float); disp = set()
    for f in glob.glob(f"gpurun_out/c3_pmc/{n}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_render_wgwf<false" not in r["Kernel_Name"]: continue   # the timed (non-counting) launches: 2 of 8 frames each
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add((f, r["Dispatch_Id"]))
    rows[n] = {k: v / 16.0 for k, v in acc.items()}   # per frame
names = sorted({k for r in rows.values() for k in r})
first = list(rows)[0]
print(f"{'counter per frame':26s} " + " ".join(f"{n:>16s}" for n in rows) + "   vs " + first)
for k in names:
    print(f"{k:26s} " + " ".join(f"{rows[n].get(k, 0):16.4g}" for n in rows) + "   " + " ".join(f"{(rows[n].get(k, 0) / rows[first][k] - 1) * 100:+7.2f} %" for n in list(rows)[1:] if rows[first].get(k)))
P
