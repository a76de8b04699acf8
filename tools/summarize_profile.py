#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into committed summaries:
   profiles/<tag>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats table (verbatim)
   profiles/<tag>_pmc.json           -- per-launch averages of the PMC counters for the render kernel(s)
   profiles/r01_pmc_traffic.json     -- HBM bytes per launch (what bench.py reports as roofline.traffic)
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB
units, collected in separate passes; on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x,
but this kernel's reads are scattered 16-byte gathers (uncalibrated width), so no correction factor is applied
and both the raw and the x2 upper bound are recorded."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel_substr = sys.argv[2] if len(sys.argv) > 2 else "pt_"
root = pathlib.Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"
dst.mkdir(exist_ok=True)

for f in glob.glob(str(src / "kt" / "*" / "*_kernel_stats.csv")):
    shutil.copy(f, dst / f"{tag}_kernel_stats.csv")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(str(src / "pmc_*" / "*" / "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kernel_substr in k and "resolve" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                           scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
out = {}
for k, cs in agg.items():
    out[k] = {"launches_sampled": max(len(v) for v in cs.values()), **meta[k],
              "counters_avg_per_launch": {c: sum(v) / len(v) for c, v in sorted(cs.items())}}
(dst / f"{tag}_pmc.json").write_text(json.dumps(out, indent=1))

main = [k for k in out if "wgwf<false" in k or "wgwfILb0" in k]
if main:
    c = out[main[0]]["counters_avg_per_launch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        (dst / "r01_pmc_traffic.json").write_text(json.dumps({
            "kernel": main[0], "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
            "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
            "hbm_bytes_per_launch": fetch + write,
            "frames_per_launch": 16,  # tools/profile.sh: every launch of the profiled command covers 16 frames
            "hbm_bytes_per_launch_upper_bound_fetch_x2": 2 * fetch + write,
            "note": "FETCH_SIZE/WRITE_SIZE are KiB; scattered 16-B gathers, no gfx950 x2 streaming correction applied"}, indent=1))
print(json.dumps(out, indent=1)[:3000])
