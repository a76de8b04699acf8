#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_phase/ (tools/gpu_phase_util.sh) -> a per-phase / per-arm table of vector-lane utilisation (printed; redirect into profiles/<tag>_lane_util.txt).
Lane utilisation = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 (calibrated 1.000 / 0.500 on full / half exec masks, profiles/r03_ubench_valu.json).
The shade phase (with top-up and bookkeeping) is the render kernel minus the replayed traverse phase; the arms of the traversal step come from the traversal-statistics
build's lane counts and the instruction counts of csrc/trav_asm.hip.h (counted from the source by this script)."""
import collections, csv, glob, json, pathlib, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = pathlib.Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}_phase"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(str(src / "pmc" / "*" / "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pt_" in k:
            rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
frames = 8
def per_frame(k, c, which):
    v = sorted(rows[k][c])
    v = [x for _, x in v]
    v = v[which] if isinstance(which, slice) else [v[which]]
    return sum(v) / len(v) / frames
render = [k for k in rows if "pt_render_wgwf" in k]
replay = [k for k in rows if "pt_replay_traverse" in k]
if not render or not replay:
    sys.exit(f"kernels not found in {src}: {list(rows)}")
R, T = render[0], replay[0]
# the render kernel's launches: warm, plain, recorded (the third one writes the log): take the second; the replay launches: all
tot = {c: per_frame(R, c, 1) for c in rows[R]}
trv = {c: per_frame(T, c, slice(None)) for c in rows[T]}
def util(d): return d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"] / 64.0
rest = {c: tot[c] - trv[c] for c in tot}
print(f"Vector-lane utilisation by phase -- headline config, {frames} frames per launch, per FRAME; {R.split('(')[0]}")
print(f"{'':34s}{'VALU insts':>14s}{'lane util':>11s}{'SALU insts':>13s}{'branches':>11s}{'SALU+br per VALU':>18s}")
for name, d in (("render kernel (all phases)", tot), ("traverse phase alone (replay)", trv), ("shade + top-up + rest (difference)", rest)):
    print(f"{name:34s}{d['SQ_INSTS_VALU'] / 1e6:12.1f} M{util(d):11.3f}{d['SQ_INSTS_SALU'] / 1e6:11.1f} M{d['SQ_INSTS_BRANCH'] / 1e6:9.1f} M{(d['SQ_INSTS_SALU'] + d['SQ_INSTS_BRANCH']) / d['SQ_INSTS_VALU']:18.3f}")
print(f"idle lane-slots (64 x insts - thread cycles), share of the kernel's: traverse {100 * (trv['SQ_INSTS_VALU'] * 64 - trv['SQ_THREAD_CYCLES_VALU']) / (tot['SQ_INSTS_VALU'] * 64 - tot['SQ_THREAD_CYCLES_VALU']):.0f} %, "
      f"shade + rest {100 * (rest['SQ_INSTS_VALU'] * 64 - rest['SQ_THREAD_CYCLES_VALU']) / (tot['SQ_INSTS_VALU'] * 64 - tot['SQ_THREAD_CYCLES_VALU']):.0f} %")
ts = src / "travstats.json"
if ts.exists():
    t = json.loads(ts.read_text())
    it, li, fk, lf, mixed = t["wave_iters"], t["lane_iters"], t["fork_lane"], t["leaf_lane"], t["mixed_iters"]
    print(f"\nThe traversal step's two arms (traversal-statistics build, {t['frames']} frames): {it / t['frames'] / 1e6:.2f} M wave-steps per frame, {li / it:.1f} of 64 lanes running;")
    print(f"  fork arm: runs in every step,                      {fk / it:5.1f} lanes enabled = {fk / it / 64:.3f}")
    print(f"  leaf arm: runs in the steps with a lane at a leaf ({100 * mixed / it:.1f} % of the steps carry both arms), {lf / it:5.1f} lanes enabled per step = {lf / it / 64:.3f}")
