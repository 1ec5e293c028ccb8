#!/usr/bin/env python3
"""ISA census of the step kernels (runs here, no GPU): compiles csrc/ois_kernels.hip to gfx950
assembly with the Makefile's flags and prints, per kernel, registers / scratch / LDS and the VALU
instruction count of the substep loop (the depth-2 loop of the fused rollout) by class.
    python tools/isa_census.py [kernel-name-substring ...]      default: step_kernelILi1ELb1ELi4
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "options_in_simulation_amd", "csrc")
FLAGS = ("-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
         "-mllvm -align-all-nofallthru-blocks=6 -Wno-unused-function -Wno-pass-failed --cuda-device-only -S").split()


def census(asm, want):
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
        blocks.append(cur)
        md = meta.get(name, {})
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


def main():
    want = sys.argv[1:] or ["step_kernelILi1ELb1ELi4"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, os.path.join(CSRC, "ois_kernels.hip")], check=True,
                       stderr=subprocess.DEVNULL)
        census(open(out).read(), want)


if __name__ == "__main__":
    main()
