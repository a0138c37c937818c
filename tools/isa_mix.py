#!/usr/bin/env python3
"""Instruction mix per kernel from a `hipcc -S --cuda-device-only` dump (dev aid)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
ops = None
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        ops = collections.Counter()
        continue
    if name and line.startswith(".Lfunc_end"):
        if pat in name:
            tot = sum(ops.values())
            valu = sum(v for k, v in ops.items() if k.startswith("v_"))
            print(name, "total", tot, "valu", valu,
                  "ds", sum(v for k, v in ops.items() if k.startswith("ds_")),
                  "salu", sum(v for k, v in ops.items() if k.startswith("s_")))
            print("   ", ops.most_common(30))
        name = None
        continue
    if name:
        s = line.strip()
        if not s or s[0] in ".;/" or s.endswith(":"):
            continue
        ops[s.split()[0]] += 1
