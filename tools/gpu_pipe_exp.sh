#!/bin/bash
# Experiment driver (GPU box): bench.py's one-launch-per-frame figure under launch-shape settings of the overlapped single-frame path.
run() { echo "== $1"; env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['one_launch_per_frame'])"; }
for v in "$@"; do run "$v"; done
