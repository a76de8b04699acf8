#!/usr/bin/env python3
"""Register / spill metadata and ISA checks of the gfx950 code object inside libglrtx.so.

    python tools/isa_report.py                 # table of every glrtx kernel: VGPRs, SGPRs, spills, scratch, LDS
    python tools/isa_report.py --check         # exit 1 if a build invariant is broken (see check())
    python tools/isa_report.py --out FILE      # also write the table (committed as profiles/<round>_codeobj_notes.txt)

Invariants checked (they protect hand-written inline asm from register-allocation changes):
  * the list-scan kernels (pt_render_wgwf<*, true>) issue `s_load_dwordx16` from one asm statement and wait for it in
    another (trav_scan: the record is fetched one step ahead of its use).  Between the load and the `s_waitcnt
    lgkmcnt(0)` that follows it no instruction may read or copy the destination registers.
  * the traversal step's pop loop (trav_step, inline asm) keeps the sentinel REF_FIN and the ref it overwrites in DIFFERENT registers
    (both start out with the same value; sharing them made the loop write back the last culled ref instead of the sentinel).
Works without a GPU (llvm-objcopy / clang-offload-bundler / llvm-readelf / llvm-objdump from /opt/rocm).
"""
from __future__ import annotations

import argparse
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
LIB = ROOT / "opengl-raytracer_amd" / "lib" / "libglrtx.so"


def extract(lib: pathlib.Path, tmp: pathlib.Path) -> pathlib.Path:
    fat, co = tmp / "fat.bin", tmp / "gfx950.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    return co


def demangle(names):
    r = subprocess.run([str(LLVM / "llvm-cxxfilt")] if (LLVM / "llvm-cxxfilt").exists() else ["c++filt"], input="\n".join(names),
                       capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else names
    return dict(zip(names, out))


def kernels(co: pathlib.Path):
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    ks, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":  # first key of a kernel entry (keys are sorted)
            cur = {}
            ks.append(cur)
        if cur is not None and k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                      "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v if k == "name" else int(v)
    ks = [k for k in ks if "glrtx" in k.get("name", "") and "rocprim" not in k["name"]]
    dm = demangle([k["name"] for k in ks])
    for k in ks:
        k["pretty"] = re.sub(r"\(.*$", "", dm[k["name"]]).replace("void ", "")
    return ks


def table(ks) -> str:
    hdr = f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}"
    rows = [hdr, "-" * len(hdr)]
    for k in ks:
        rows.append(f"{k['pretty'][:58]:58s} {k['vgpr_count']:5d} {k['agpr_count']:5d} {k['sgpr_count']:5d} {k['vgpr_spill_count']:6d} "
                    f"{k['sgpr_spill_count']:6d} {k['private_segment_fixed_size']:7d} {k['group_segment_fixed_size']:6d}")
    return "\n".join(rows)


def is_list_scan(pretty: str) -> bool:
    """pt_render_wgwf<COUNT_RAYS, VINE, PAIR>: the second template argument selects the list scan of a vine tree."""
    m = re.search(r"pt_render_wgwf<([^>]*)>", pretty)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(",")]
    return len(args) > 1 and args[1] == "true"


def sreg_set(tok: str):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_scalar_prefetch(co: pathlib.Path, ks) -> list:
    """No use of an s_load_dwordx16 destination between the load and the next s_waitcnt lgkmcnt(0)."""
    problems = []
    dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    body, name = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name = m.group(1)
            body[name] = []
        elif name and ln.strip():
            body[name].append(ln.split("//")[0].strip())
    for k in ks:
        if not is_list_scan(k["pretty"]):
            continue
        ins = body.get(k["name"], [])
        n_loads = 0
        pending = None  # (registers, line)
        for i, s in enumerate(ins):
            toks = re.findall(r"s\[\d+:\d+\]|s\d+", s)
            if s.startswith("s_load_dwordx16"):
                n_loads += 1
                pending = (sreg_set(toks[0]), i)
                continue
            if pending and s.startswith("s_waitcnt") and "lgkmcnt(0)" in s:
                pending = None
                continue
            if pending:
                used = set().union(*[sreg_set(t) for t in toks]) if toks else set()
                if used & pending[0]:
                    problems.append(f"{k['pretty']}: `{s}` touches the destination of the s_load_dwordx16 at instruction {pending[1]} before its s_waitcnt")
                if s.startswith(("s_branch", "s_cbranch", "s_endpgm")):
                    pending = None  # the wait is in the successor block: followed no further (the loop keeps load and wait in one block)
        if n_loads == 0:
            problems.append(f"{k['pretty']}: no s_load_dwordx16 found (the list scan no longer uses the scalar cache?)")
    return problems


def check_pop_loop(co: pathlib.Path, ks) -> list:
    """Every `v_cmp_ne_u32 vcc, 0, vSP ; v_cndmask_b32 vREF, vFIN, vREF, vcc` pair of the render kernels: vFIN != vREF, and the loop is there."""
    problems = []
    dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    body, name = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name = m.group(1)
            body[name] = []
        elif name and ln.strip():
            body[name].append(ln.split("//")[0].strip())
    for k in ks:
        if not any(t in k["pretty"] for t in ("pt_render_wgwf", "pt_render_persistent", "pt_render_kernel")):
            continue
        if is_list_scan(k["pretty"]):
            continue  # list-scan instantiations never pop
        ins = body.get(k["name"], [])
        n = 0
        for a, b in zip(ins, ins[1:]):
            if re.fullmatch(r"v_cmp_ne_u32_e32 vcc, 0, v\d+", a):
                m = re.fullmatch(r"v_cndmask_b32_e32 (v\d+), (v\d+), (v\d+), vcc", b)
                if m and m.group(1) == m.group(3):
                    n += 1
                    if m.group(2) == m.group(1):
                        problems.append(f"{k['pretty']}: pop loop: `{b}` -- the sentinel shares the ref's register")
        if n == 0:
            problems.append(f"{k['pretty']}: no pop loop found (v_cmp_ne_u32 vcc, 0, sp / v_cndmask ref, fin, ref)")
    return problems


def check_staged_waits(co: pathlib.Path, ks) -> list:
    """The hand-written traversal step of the wavefront kernel.  Behind every node fetch (two dwordx4 + two dwordx3 loads, in either order) the waits come in
    stages -- pair-cooperative fetch (round 4): s_waitcnt vmcnt(2) then vmcnt(0); one record per lane (rounds 2-3): vmcnt(3), (2), (1), (0) -- with no other
    vector-memory instruction in between (a load or store slipped into the sequence would make the counts wait for the wrong thing).  And the DPP moves of the
    pair exchange keep gfx950's manual hazards: a DPP source register is not written by the two instructions in front of it, and no v_cmpx (a vector write
    of exec) sits in the five instructions in front of a DPP instruction (the assembler inserts no wait states into inline asm)."""
    problems = []
    dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    body, name = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name = m.group(1)
            body[name] = []
        elif name and ln.strip():
            body[name].append(ln.split("//")[0].strip())

    def vregs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    for k in ks:
        if "pt_render_wgwf" not in k["pretty"] or is_list_scan(k["pretty"]):
            continue  # (list-scan instantiations have no stepping block)
        ins = body.get(k["name"], [])
        fetches = 0
        for i, s in enumerate(ins):
            if i < 3 or not all(t.startswith("global_load_dwordx") for t in ins[i - 3:i + 1]):
                continue
            widths = sorted(t.split()[0][-1] for t in ins[i - 3:i + 1])
            if widths != ["3", "3", "4", "4"] or (i + 1 < len(ins) and ins[i + 1].startswith("global_load")):
                continue
            fetches += 1
            pair = ins[i - 2].startswith("global_load_dwordx4")  # x4, x4, x3, x3: pair-cooperative; x4, x4, x3, x3 with one address register: per lane
            pair = pair and len({t.split(",")[1].strip() for t in ins[i - 3:i + 1]}) == 2  # two address registers
            stages = [2, 0] if pair else [3, 2, 1, 0]
            for t in ins[i + 1:i + 120]:
                if t.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    problems.append(f"{k['pretty']}: `{t}` inside the staged waits of the node fetch at instruction {i}")
                    break
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
                if m:
                    if int(m.group(1)) != stages[0]:
                        problems.append(f"{k['pretty']}: node fetch at instruction {i}: expected s_waitcnt vmcnt({stages[0]}), found `{t}`")
                        break
                    stages.pop(0)
                    if not stages:
                        break
            else:
                problems.append(f"{k['pretty']}: node fetch at instruction {i}: the staged waits are incomplete")
        if fetches == 0:
            problems.append(f"{k['pretty']}: no node fetch (2 x dwordx4 + 2 x dwordx3) found")
        for i, s in enumerate(ins):
            if "quad_perm" not in s:
                continue
            ops = [t.strip().rstrip(",") for t in s.split()[1:3]]
            src = vregs(ops[1]) if len(ops) > 1 else set()
            for t in ins[max(0, i - 2):i]:
                if t.startswith("v_") and not t.startswith("v_cmp"):
                    dst = vregs(t.split()[1].rstrip(","))
                    if dst & src:
                        problems.append(f"{k['pretty']}: `{s}` at instruction {i} reads a register `{t}` wrote fewer than two wait states earlier")
            for t in ins[max(0, i - 5):i]:
                if t.startswith("v_cmpx"):
                    problems.append(f"{k['pretty']}: `{s}` at instruction {i} follows `{t}` (a vector write of exec) by fewer than five wait states")
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=str(LIB))
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        co = extract(pathlib.Path(a.lib), pathlib.Path(d))
        ks = kernels(co)
        t = table(ks)
        print(t)
        if a.out:
            pathlib.Path(a.out).write_text("llvm-readelf --notes of the gfx950 code object in libglrtx.so (tools/isa_report.py)\n\n" + t + "\n")
        if a.check:
            bad = check_scalar_prefetch(co, ks) + check_pop_loop(co, ks) + check_staged_waits(co, ks)
            for b in bad:
                print("ISA CHECK FAILED:", b)
            sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
