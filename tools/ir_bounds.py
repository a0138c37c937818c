#!/usr/bin/env python3
"""Worst-case magnitude of every multiply operand of the 64-point lifting network (dev aid).

The network is linear up to rounding, so |register| <= L1(row of the linearised network) *
max|input| + accumulated rounding (each RSH1 / multiply-shift adds at most 1, propagated
with the same gains; bounded here by a generous slack).  Prints, for the column pass
(|input| <= 23 100) and the row pass (|input| <= 8 * 23 100), how many
(a*K + R) >> S steps can overflow int32 and how many qualify for the high-dword form.
usage: ir_bounds.py [tools/ir/fdct64_ir.json]
"""
import json
import sys

import numpy as np


def analyse(path):
    ir = json.load(open(path))
    n = ir["n_regs"]
    lin = np.zeros((n, ir["n_in"]))
    err = np.zeros(n)                      # accumulated rounding error bound
    for i in range(ir["n_in"]):
        lin[i, i] = 1.0
    mul_ops = []
    for op in ir["ops"]:
        k = op[0]
        if k == "SUB":
            d, a, b = op[1:]
            lin[d], err[d] = lin[a] - lin[b], err[a] + err[b]
        elif k == "ADD":
            d, a, b = op[1:]
            lin[d], err[d] = lin[a] + lin[b], err[a] + err[b]
        elif k == "RSH1":
            d, a = op[1:]
            lin[d], err[d] = lin[a] * 0.5, err[a] * 0.5 + 1
        elif k in ("MLA", "MLS"):
            d, a, K, R, S = op[1:]
            mul_ops.append((np.abs(lin[a]).sum(), err[a], K, R, S))
            sgn = 1.0 if k == "MLA" else -1.0
            lin[d], err[d] = lin[d] + sgn * lin[a] * (K / 2.0 ** S), err[d] + err[a] * K / 2.0 ** S + 1
        elif k == "NEG":
            d, a = op[1:]
            lin[d], err[d] = -lin[a], err[a]
        elif k == "MOV":
            d, a = op[1:]
            lin[d], err[d] = lin[a].copy(), err[a]
        else:
            raise SystemExit("unknown op " + k)
    out_l1 = max(np.abs(lin[r]).sum() for r in ir["out_regs"])
    return mul_ops, out_l1


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "tools/ir/fdct64_ir.json"
    mul_ops, out_l1 = analyse(path)
    print("multiplies:", len(mul_ops), " max output L1 gain: %.3f" % out_l1)
    for name, bin_ in (("column pass", 23100.0), ("row pass", 23100.0 * out_l1 + 64)):
        safe = [(l1 * bin_ + e) * K + R < 2 ** 31 for (l1, e, K, R, S) in mul_ops]
        hi = [s and K < (1 << (S - 1)) for s, (l1, e, K, R, S) in zip(safe, mul_ops)]
        print("%-12s |in| <= %8.0f : %3d of %d cannot overflow int32, %3d of those have K < 2^(S-1)"
              % (name, bin_, sum(safe), len(mul_ops), sum(hi)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
