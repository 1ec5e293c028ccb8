#!/usr/bin/env python3
"""ISA census of the step kernels (runs here, no GPU): compiles csrc/ois_kernels.hip to gfx950
assembly with the Makefile's flags and prints, per kernel, registers / scratch / LDS and the VALU
instruction count of the substep loop (the depth-2 loop of the fused rollout) by class.
    python tools/isa_census.py [kernel-name-substring ...]      default: step_kernelILi1ELb1ELi4
    python tools/isa_census.py --json profiles/r4_isa_census.json      every step-kernel instantiation, keyed by the id
                                                                         ois_kernel_info reports, with the kernel-source hash,
                                                                         + the phase table of the headline instantiation
    python tools/isa_census.py --phases [kernel-name-substring]        where the instructions of ONE policy step are: every
        instruction of the kernel attributed — through the line table (-gline-tables-only: same code) and its inlined-at
        chain — to the phase of step_body it was written in (the `// [phase: ...]` markers of csrc/ois_kernels.hip), the
        substep loop apart; regions the bench configuration does not execute (`// [census: not in the bench configuration]`,
        phases marked rare / only) listed but not summed
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "options_in_simulation_amd", "csrc")
FLAGS = ("-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
         "-mllvm -align-all-nofallthru-blocks=6 -Wno-unused-function -Wno-pass-failed --cuda-device-only -S").split()


def kernel_id(mangled):
    """_ZN..11step_kernelILi1ELb1ELi4ELb1EE... -> step_kernel<1,1,4,1> (the id of ois_kernel_info)"""
    m = re.search(r"\d+(step_kernel\w*?)I((?:L[ib]\d+E)+)E", mangled)
    if not m:
        return None
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    # the last template argument of the three shells is FOLD (continuation launches of a level-curriculum rollout): ois_kernel_info
    # names the shell an ois_step launches, i.e. without it
    # step_kernel_terrain<PK, NIT, FOLD, TSTRIDE>: the last one is ois_terrain.footprint_stride ("+stride2", as ois_kernel_info names it)
    stride = args.pop() if m.group(1) == "step_kernel_terrain" and len(args) == 4 else "1"
    fold = args.pop() if len(args) == {"step_kernel": 5, "step_kernel_w2": 4, "step_kernel_terrain": 3}.get(m.group(1), -1) else "0"
    return m.group(1) + "<" + ",".join(args) + ">" + ("+stride2" if stride == "2" else "") + ("+fold" if fold == "1" else "")


def census(asm, want, quiet=False):
    out = {}
    lines = asm.split("\n")
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\w+:.*\n)+)", asm):
        meta[m.group(1)] = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
    for i, l in enumerate(lines):
        m0 = re.match(r"^(_Z\w+):", l)
        if not (m0 and any(w in m0.group(1) for w in want)):
            continue
        name = m0.group(1)
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        blocks, cur = [], ["entry", "", []]
        for b in lines[i + 1:end]:
            s = b.strip()
            mm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
            if mm:
                blocks.append(cur)
                cur = [mm.group(1), mm.group(2) or "", []]
                continue
            if s.startswith(";") and not cur[2]:
                cur[1] += " " + s  # loop annotations continue on comment lines after the label
            if not s or s[0] in ";.":
                continue
            cur[2].append(s.split()[0])
            if s.split()[0] == "s_cbranch_execz":
                # the fall-through of an execz branch runs only while some lane took the `if`: its own pseudo-block, so that a rare
                # path that the assembler printed inside the label of the main path is not counted as executed
                blocks.append(cur)
                cur = [cur[0] + "+execz", cur[1], []]
        blocks.append(cur)
        md = meta.get(name, {})
        if quiet:
            inner = [b for b in blocks if "Depth=2" in b[1]]
            # the rare branches of the loop (velocity cap, exact quaternion normalisation: short blocks with a sqrt) are
            # not on the executed path
            main = [b for b in inner if not (len(b[2]) < 100 and any(re.match(r"v_(sqrt|rsq|div|rcp)", i) for i in b[2]))]
            ins_all = [i for b in main for i in b[2]]
            valu = [i for i in ins_all if i.startswith("v_")]
            flop = sum(4 if i.startswith("v_pk_fma") else 2 if i.startswith("v_pk_") and not i.startswith("v_pk_mov") else
                       2 if re.match(r"v_(fma|fmac|fmaak|fmamk|mad|mac)_f32", i) else 1 if re.match(r"v_(add|sub|subrev|mul)_f32", i) else 0
                       for i in ins_all)
            out[kernel_id(name)] = {"vgpr": int(md.get("vgpr_count", 0)), "scratch_bytes": int(md.get("private_segment_fixed_size", 0)),
                                    "lds_bytes": int(md.get("group_segment_fixed_size", 0)),
                                    "substep_valu": len(valu), "substep_valu_packed": sum(1 for i in valu if i.startswith("v_pk_")),
                                    "substep_flop_per_lane": flop, "substep_blocks": [len(b[2]) for b in inner], "substep_blocks_counted": len(main),
                                    # footprint stride 2: the loop body is a PAIR of substeps (odd + even), the counts above are the pair's
                                    "loop_substeps": 2 if "+stride2" in kernel_id(name) else 1,
                                    "whole_kernel_instructions": sum(len(b[2]) for b in blocks)}
            continue
        print(f"== {name[:100]}\n   vgpr {md.get('vgpr_count')} agpr {md.get('agpr_count')} sgpr {md.get('sgpr_count')} "
              f"scratch {md.get('private_segment_fixed_size')} B  lds {md.get('group_segment_fixed_size')} B")
        inner = [b for b in blocks if "Depth=2" in b[1]]
        tot = {}
        for b in inner:
            for ins in b[2]:
                cls = ("pk" if ins.startswith("v_pk_") else "med3/minmax" if re.match(r"v_(med3|max|min)", ins) else
                       "fma/mul/add" if re.match(r"v_(fma|mul|add|sub|fmac|mac)", ins) else "mov/cndmask" if re.match(r"v_(mov|cndmask|accvgpr)", ins)
                       else "other valu" if ins.startswith("v_") else "salu" if ins.startswith("s_") else "mem")
                tot[cls] = tot.get(cls, 0) + 1
        flop = 0
        for b in inner:
            for ins in b[2]:
                flop += (4 if ins.startswith("v_pk_fma") else 2 if ins.startswith("v_pk_") and not ins.startswith("v_pk_mov") else
                         2 if re.match(r"v_(fma|fmac|fmaak|fmamk|mad|mac)_f32", ins) else 1 if re.match(r"v_(add|sub|subrev|mul)_f32", ins) else 0)
        valu = sum(v for k, v in tot.items() if k not in ("salu", "mem"))
        print(f"   substep loop FLOP per lane (fma = 2, packed = x2; all blocks incl. the rare velocity-cap branch): {flop}")
        print(f"   substep loop ({len(inner)} blocks): VALU {valu}  " + "  ".join(f"{k} {v}" for k, v in sorted(tot.items())))
        big = max(blocks, key=lambda b: len(b[2]))
        print(f"   largest block {big[0]}: {len(big[2])} instr; whole kernel {sum(len(b[2]) for b in blocks)} instr")
    return out


def phase_table(want="step_kernelILi1ELb1ELi4ELb1ELb0"):
    """{phase: {...}} for the first kernel whose mangled name contains `want` + the executed totals of a launch of k steps"""
    src_path = os.path.join(CSRC, "ois_kernels.hip")
    src = open(src_path).read().split("\n")
    phases, skips, cur_skip = [], [], None  # (first line, name); (first line, last line, why)
    body0 = next(i for i, l in enumerate(src, 1) if "OIS_DEV void step_body(" in l)
    body1 = next(i for i, l in enumerate(src, 1) if i > body0 and l.startswith("// Kernel shells around the same body"))
    for i, l in enumerate(src, 1):
        m = re.search(r"// \[phase: (.*)\]\s*$", l)
        if m and body0 < i < body1:
            phases.append((i, m.group(1)))
        m = re.search(r"// \[census: not in the bench configuration: (.*)\]", l)
        if m:
            cur_skip = (i, m.group(1))
        if "// [census: end]" in l and cur_skip:
            skips.append((cur_skip[0], i, cur_skip[1])); cur_skip = None
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-gline-tables-only", "-o", out, src_path], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read().split("\n")
    i0 = next(i for i, l in enumerate(asm) if re.match(r"^_Z\w+:", l) and want in l)
    i1 = next(j for j in range(i0, len(asm)) if asm[j].startswith(".Lfunc_end"))
    name = asm[i0].split(":")[0]
    table, depth2, frames = {}, False, []
    rng_lines = None
    dev = open(os.path.join(CSRC, "ois_device.h")).read().split("\n")
    r0 = next(i for i, l in enumerate(dev, 1) if "OIS_DEV void philox4x32_10" in l)
    r1 = next(i for i, l in enumerate(dev, 1) if i > r0 and "OIS_DEV void draw4" in l) + 8
    for l in asm[i0 + 1:i1]:
        s_ = l.strip()
        if re.match(r"^(\.LBB\d+_\d+):", s_) or s_.startswith("; %bb."):
            depth2 = "Depth=2" in s_
            continue
        if s_.startswith(";") and "Loop Header: Depth=" in s_:
            depth2 = "Depth=2" in s_
        m = re.match(r"^\.loc\s+\d+\s+\d+.*?;\s*(.*)$", s_)
        if m:
            frames = [(os.path.basename(a), int(b)) for a, b in re.findall(r"([^\s\[\]@:]+):(\d+)(?::\d+)?", m.group(1))]
            continue
        if not s_ or s_[0] in ";.":
            continue
        op = s_.split()[0]
        line = next((ln for f, ln in frames if f == "ois_kernels.hip" and body0 <= ln < body1), None)
        if line is None:
            ph = "(unattributed)"
        else:
            ph = next((n for a, n in reversed(phases) if a <= line), "(before the first marker)")
            sk = next((w for a, b, w in skips if a <= line <= b), None)
            if sk:
                ph = "not in the bench configuration: " + sk
        if ph.startswith("substeps"):
            ph = "substep loop body (x nsub - 1)" if depth2 else "first substep (exact reciprocals) + body <-> world glue (per step)"
        elif frames and frames[0][0] == "ois_device.h" and r0 <= frames[0][1] <= r1 and not ph.startswith(("not in", "reset_idx")):
            ph = "Philox4x32-10 of the switch draw (per step)"
        t = table.setdefault(ph, {"instructions": 0, "valu": 0, "valu_packed": 0, "lane_spill": 0, "v_mov": 0, "salu": 0, "mem": 0})
        t["instructions"] += 1
        if op.startswith("v_"):
            t["valu"] += 1
            t["valu_packed"] += op.startswith("v_pk_")
            t["lane_spill"] += op in ("v_readlane_b32", "v_writelane_b32")
            t["v_mov"] += op.startswith(("v_mov", "v_accvgpr"))
        elif op.startswith("s_"):
            t["salu"] += 1
        else:
            t["mem"] += 1
    per_launch = sum(v["valu"] for k, v in table.items() if "(per launch)" in k or "last step of a launch" in k)
    per_step = sum(v["valu"] for k, v in table.items() if "(per step)" in k)
    loop = table.get("substep loop body (x nsub - 1)", {}).get("valu", 0)
    return {"kernel": kernel_id(name), "phases": table,
            "valu_static": {"per_launch": per_launch, "per_step_outside_the_loop": per_step, "substep_loop_body_all_blocks": loop},
            "note": "static counts of the instructions WRITTEN in each phase (all blocks of it, incl. branches a step does not take: the loop body's "
                    "executed path is `substep_valu` above); executed VALU per wave of a k-step launch ~ per_launch + k * (per_step + 19 * substep_valu)"}


def main():
    js = None
    args = sys.argv[1:]
    if args[:1] == ["--phases"]:
        res = phase_table(*(args[1:2] or []))
        print(f"== {res['kernel']}")
        for k, v in sorted(res["phases"].items(), key=lambda kv: -kv[1]["valu"]):
            print(f"   {v['valu']:5d} VALU ({v['valu_packed']:3d} packed, {v['lane_spill']:3d} lane spill, {v['v_mov']:3d} mov) {v['salu']:4d} SALU {v['mem']:3d} mem   {k}")
        print("  ", res["valu_static"])
        return
    if args[:1] == ["--json"]:
        js, args = args[1], ["step_kernel"]
    want = args or ["step_kernelILi1ELb1ELi4"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, os.path.join(CSRC, "ois_kernels.hip")], check=True,
                       stderr=subprocess.DEVNULL)
        res = census(open(out).read(), want, quiet=js is not None)
    if js:
        import json
        sys.path.insert(0, ROOT)
        import bench
        doc = {"kernel_src_sha256": bench.kernel_source_hash(), "lanes_per_substep_note": "per LANE and substep, the executed path of the substep loop "
               "(the short rare blocks with a sqrt / division — velocity cap, exact quaternion normalisation — are not counted); with lanes_per_arena = 2 an arena is two lanes; flop: fma = 2, packed = x2",
               "kernels": res, "phases_of_the_headline_kernel": phase_table()}
        json.dump(doc, open(js, "w"), indent=1, sort_keys=True)
        print("wrote", js, len(res), "kernels")


if __name__ == "__main__":
    main()
