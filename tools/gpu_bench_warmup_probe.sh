mkdir -p gpurun_out/s3
for w in 5 45 100 5; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup $w --no-cpu-baseline --no-single > gpurun_out/s3/bench_w$w.json 2> gpurun_out/s3/bench_w$w.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/s3/bench_w$w.json").read().strip().splitlines()[-1])
print("warmup $w:", d["value"], d["ms_per_step"], d["config"].get("event_ms_per_step"), d["roofline"]["kernel_ms_avg"])
PY
done
