#!/usr/bin/env python3
"""ISA census of the step kernels (runs here, no GPU): compiles csrc/ois_kernels.hip to gfx950
assembly with the Makefile's flags and prints, per kernel, registers / scratch / LDS and the VALU
instruction count of the substep loop (the depth-2 loop of the fused rollout) by class.
    python tools/isa_census.py [kernel-name-substring ...]      default: step_kernelILi1ELb1ELi4
    python tools/isa_census.py --json profiles/r4_isa_census.json      every step-kernel instantiation, keyed by the id
                                                                         ois_kernel_info reports, with the kernel-source hash
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
    fold = args.pop() if len(args) == {"step_kernel": 5, "step_kernel_w2": 4, "step_kernel_terrain": 3}.get(m.group(1), -1) else "0"
    return m.group(1) + "<" + ",".join(args) + ">" + ("+fold" if fold == "1" else "")


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


def main():
    js = None
    args = sys.argv[1:]
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
               "kernels": res}
        json.dump(doc, open(js, "w"), indent=1, sort_keys=True)
        print("wrote", js, len(res), "kernels")


if __name__ == "__main__":
    main()
