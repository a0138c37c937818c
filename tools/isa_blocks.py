#!/usr/bin/env python3
"""Per-basic-block instruction classes of one kernel in a hipcc -S dump (dev aid).

usage: isa_blocks.py file.s kernel-substring
Classes follow tools/microbench/opbench.hip (gfx950, 2 waves/SIMD): 'cheap' VALU issue in
one pass (VOP1/VOP2 e32 add/sub/and/or/mov/lshr/ashr, f32 mul/add/fma), 'exp' = everything
else on the VALU (multiplies, VOP3 integer, SDWA/DPP, lshl, cmp, cvt, packed).
"""
import re
import sys

CHEAP = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
         "v_mov_b32_e32", "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_mul_f32_e32", "v_add_f32_e32",
         "v_sub_f32_e32", "v_fma_f32", "v_fmac_f32_e32", "v_not_b32_e32"}


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    inside = False
    blocks = []
    cur = None
    for ln, line in enumerate(txt, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = pat in m.group(1)
            if inside:
                cur = {"name": "entry", "line": ln, "cheap": 0, "exp": 0, "salu": 0, "ds": 0, "vmem": 0, "br": []}
                blocks.append(cur)
            continue
        if not inside:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            cur = {"name": m.group(1), "line": ln, "cheap": 0, "exp": 0, "salu": 0, "ds": 0, "vmem": 0, "br": []}
            blocks.append(cur)
            continue
        s = line.strip()
        if not s or s[0] in ".;/":
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["cheap" if op in CHEAP else "exp"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                cur["br"].append(op[2:] + "->" + s.split()[-1])
        elif op.startswith("ds_"):
            cur["ds"] += 1
        else:
            cur["vmem"] += 1
    for b in blocks:
        tot = b["cheap"] + b["exp"] + b["salu"] + b["ds"] + b["vmem"]
        if tot:
            print("%-10s @%5d  cheap %4d exp %4d salu %4d ds %3d vmem %3d  %s" %
                  (b["name"], b["line"], b["cheap"], b["exp"], b["salu"], b["ds"], b["vmem"], " ".join(b["br"])))


main()
