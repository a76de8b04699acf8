#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into committed summaries:
   profiles/<tag>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats table (verbatim)
   profiles/<tag>_pmc.json           -- per-launch averages of the PMC counters for the render kernel(s)
   profiles/<tag>_pmc_summary.json   -- per-FRAME figures bench.py reports: HBM bytes (roofline.traffic), VALU instructions, lane utilisation
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB
units, collected in separate passes; on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x.
The kernel's HBM-side reads are dominated by unit-stride float4 streams (path state, ray records), so the x2
correction is applied; the uncorrected sum is recorded next to it.
Usage: summarize_profile.py <tag> [kernel substring] [frames per launch of the profiled command]"""
import collections
import csv
import statistics
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

def newest(pattern):
    """Per pass directory the files of the NEWEST run only: gpurun merges a call's output into the local gpurun_out/, where an earlier call's files remain."""
    by_dir = collections.defaultdict(list)
    for f in glob.glob(pattern):
        by_dir[pathlib.Path(f).parent].append(f)
    return [max(fs, key=lambda f: pathlib.Path(f).stat().st_mtime) for fs in by_dir.values()]


for f in newest(str(src / "kt" / "*" / "*_kernel_stats.csv")):
    shutil.copy(f, dst / f"{tag}_kernel_stats.csv")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in newest(str(src / "pmc_*" / "*" / "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kernel_substr in k and "resolve" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            # rocprofv3's VGPR_Count is the architected half of the unified register file as the dispatch packet states it (64 for a kernel that
            # uses 126): recorded under its own name, never as "the kernel's VGPRs" -- those come from the code object (below)
            meta[k] = dict(rocprof_vgpr_count_field=int(r["VGPR_Count"]), rocprof_accum_vgpr_count_field=int(r.get("Accum_VGPR_Count", 0) or 0),
                           rocprof_sgpr_count_field=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                           scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                # the clock the kernel ran at: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                dur_ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                if dur_ns > 0:
                    agg[k]["_clock_ghz"].append(float(r["Counter_Value"]) / 8.0 / dur_ns)
                    agg[k]["_grbm_pass_duration_ms"].append(dur_ns * 1e-6)
out = {}
if not agg and (dst / f"{tag}_pmc.json").exists():  # re-summarise a committed table
    out = json.loads((dst / f"{tag}_pmc.json").read_text())
for k, cs in agg.items():
    out[k] = {"launches_sampled": max(len(v) for v in cs.values()), **meta[k],
              "counters_avg_per_launch": {c: sum(v) / len(v) for c, v in sorted(cs.items()) if not c.startswith("_")}}
    if "_clock_ghz" in cs:  # median over the launches of the GRBM pass (the first launch of a process runs at a lower clock)
        out[k]["clock_ghz"] = statistics.median(cs["_clock_ghz"])
        out[k]["clock_ghz_per_launch"] = [round(v, 4) for v in cs["_clock_ghz"]]
        out[k]["grbm_pass_duration_ms"] = statistics.median(cs["_grbm_pass_duration_ms"])
if agg:
    (dst / f"{tag}_pmc.json").write_text(json.dumps(out, indent=1))

def code_object_registers(kernel_name):
    """vgpr / sgpr / spills of the kernel from the gfx950 code object in the built library (tools/isa_report.py), not from rocprofv3's fields."""
    try:
        import importlib.util, tempfile
        spec = importlib.util.spec_from_file_location("isa_report", root / "tools" / "isa_report.py")
        ir = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ir)
        with tempfile.TemporaryDirectory() as td:
            ks = ir.kernels(ir.extract(ir.LIB, pathlib.Path(td)))
        short = kernel_name.split("(")[0].replace("void ", "").strip()
        for k in ks:
            if k.get("pretty", "").strip() == short:
                return {"vgpr": k.get("vgpr_count"), "agpr": k.get("agpr_count"), "sgpr": k.get("sgpr_count"), "vgpr_spills": k.get("vgpr_spill_count"),
                        "sgpr_spills": k.get("sgpr_spill_count"), "scratch": k.get("private_segment_fixed_size"), "registers_from": "code object (llvm-readelf --notes)"}
    except Exception as e:  # summaries are still written
        return {"registers_from": f"unavailable: {type(e).__name__}: {e}"}
    return {"registers_from": "kernel not found in the code object"}


main = [k for k in out if "wgwf<false" in k or "wgwfILb0" in k]
if main:
    FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 16  # frames per launch of the profiled command (tools/profile.sh)
    m = out[main[0]]
    c = m["counters_avg_per_launch"]
    import hashlib
    hh = hashlib.sha256()
    pkg = root / "opengl-raytracer_amd"
    for f in sorted((pkg / "csrc").glob("*")) + [pkg / "Makefile"]:  # == bench.py: kernel_source_sha16()
        hh.update(f.name.encode())
        hh.update(f.read_bytes())
    summ = {"kernel": main[0], "source": f"profiles/{tag}_pmc.json", "frames_per_launch": FRAMES, "kernel_source_sha16": hh.hexdigest()[:16],
            **code_object_registers(main[0])}
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
        # cycles the 256 CUs' address / L1 front ends had work, over the kernel's cycles: reads 0.965-0.98 on the saturated micro-benchmark (profiles/r04_ta_counters.json)
        summ["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
    if "grbm_pass_duration_ms" in m:
        summ["kernel_ms_per_frame_profiled"] = m["grbm_pass_duration_ms"] / FRAMES  # duration of the dispatches the clock and the TA counter were read from
    if "clock_ghz" in m:
        summ["clock_ghz"] = m["clock_ghz"]
        summ["clock_note"] = ("GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same dispatch (rocprofv3 --pmc GRBM_GUI_ACTIVE pass, median over its launches): the clock the "
                              "issue rooflines of bench.py are priced at")
    # node fetches of the traverse phase: wave-steps x 4 load instructions, from the traversal-statistics build run by tools/profile.sh
    ts = src / "travstats.json"
    if not ts.exists() and (dst / f"{tag}_travstats.json").exists():
        ts = dst / f"{tag}_travstats.json"
    if ts.exists():
        t = json.loads(ts.read_text())
        shutil.copy(ts, dst / f"{tag}_travstats.json") if ts.parent != dst else None
    if ts.exists() and t.get("wave_iters", 0) > 0:  # (the list scan of a vine tree has no stepping block: no node fetches)
        summ["wave_steps_per_frame"] = t["wave_iters"] / t["frames"]
        summ["node_fetch_insts_per_frame"] = 4.0 * t["wave_iters"] / t["frames"]
        summ["lanes_per_wave_step"] = t["lane_iters"] / t["wave_iters"]
        summ["lines_per_wave_step"] = t["lines"] / t["wave_iters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        summ.update(fetch_bytes_per_frame_raw=fetch / FRAMES, write_bytes_per_frame=write / FRAMES,
                    hbm_bytes_per_frame=(2 * fetch + write) / FRAMES, hbm_bytes_per_frame_uncorrected=(fetch + write) / FRAMES,
                    traffic_note="KiB units; FETCH_SIZE doubled (gfx950 counts a 128-B read request as 64 B: MI355X_MICROARCH.md, HBM) -- exact for "
                                 "the unit-stride float4 state/ray-record streams that dominate, an upper bound for the scattered node gathers; "
                                 "Infinity-Cache hits are included in both counters")
    if "SQ_INSTS_VALU" in c:
        summ["valu_insts_per_frame"] = c["SQ_INSTS_VALU"] / FRAMES
    for key, name in (("SQ_INSTS_SALU", "salu_insts_per_frame"), ("SQ_INSTS_BRANCH", "branch_insts_per_frame"), ("SQ_INSTS_LDS", "lds_insts_per_frame"),
                      ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_frame"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts_per_frame")):
        if key in c:
            summ[name] = c[key] / FRAMES
    if "SQ_INSTS_VMEM_RD" in c:
        summ["vmem_insts_per_frame"] = (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0.0)) / FRAMES
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_INSTS_VALU" in c:
        # enabled lanes per vector instruction / 64.  Calibrated with tools/ubench/valu.hip (profiles/r03_ubench_valu.json): 1.000 with all
        # lanes enabled, 0.500 with exec = 0xFFFFFFFF; SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU there and is not used
        summ["lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64.0
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        summ["wave_cycles_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        summ["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    (dst / f"{tag}_pmc_summary.json").write_text(json.dumps(summ, indent=1))
    print(json.dumps(summ, indent=1))
