#!/usr/bin/env python3
"""A/B of two builds of libffv2amd.so on the SAME GPU box (box-to-box spread is +-2 %, more than most
single changes): alternates `bench.py --no-cpu-baseline` between the libraries and prints the T-stage
kernel time of every run.  usage: tools/ab_bench.py A.so B.so [rounds]
The shipped library is swapped by file copy, so run this only on a scratch copy (the GPU box)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "ffmpeg_ffv2_amd", "libffv2amd.so")


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    keep = SO + ".keep"
    shutil.copy(SO, keep)
    res = {a: [], b: []}
    try:
        for _ in range(rounds):
            for lib in (a, b):
                shutil.copy(lib, SO)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"],
                                     capture_output=True, text=True, check=True).stdout
                d = json.loads(out.strip().splitlines()[-1])
                res[lib].append((d["roofline"]["kernel_ms_avg"], d["ms_per_step"]))
                print(os.path.basename(lib), res[lib][-1], flush=True)
    finally:
        shutil.copy(keep, SO)
        os.remove(keep)
    for lib in (a, b):
        k = sorted(x[0] for x in res[lib])
        print("%-28s T-stage median %.4f ms (min %.4f)" % (os.path.basename(lib), k[len(k) // 2], k[0]))


if __name__ == "__main__":
    main()
