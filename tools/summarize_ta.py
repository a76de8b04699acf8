#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_ta/ (tools/profile_ta.sh) -> profiles/<tag>_ta_counters.json: per-launch averages of the TA / TCP / TD counters for the render
kernel (pass directories) and for the TA micro-benchmark at saturation (cal_* directories), with the launch duration of the same dispatches."""
import collections, csv, glob, json, os, pathlib, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = pathlib.Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}_ta"
out = collections.OrderedDict()
for d in sorted(glob.glob(str(src / "*") + "/")):
    name = os.path.basename(os.path.dirname(d))
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wgwf<false" in k or "ta_kernel" in k:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    agg["_duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if agg:
        out[name] = {c: sum(v) / len(v) for c, v in agg.items()}
        out[name]["_launches"] = len(agg["_duration_ns"])
(root / "profiles").mkdir(exist_ok=True)
(root / "profiles" / f"{tag}_ta_counters.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
