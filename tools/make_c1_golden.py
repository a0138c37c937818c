#!/usr/bin/env python3
"""Writes tests/golden/c1_packets.json: per-frame size + md5 of BASELINE config 1 as it reaches
encode2() (320x240 planar 4:4:4 8-bit, 30 synthetic frames S1/S2 alternating, qp=0), produced by the
CPU oracle (which is pinned by the reference's known answers, tests/test_oracle_kat.py).  The frames are
regenerated from seeds, only the digests are committed."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib
from ffmpeg_ffv2_amd import frames as synth
o = oracle_lib.load()
out = []
for n in range(30):
    fr = synth.make("S1" if n % 2 == 0 else "S2", n, 3, 240, 320, 8)
    pk = o.encode(fr, "yuv444p")
    out.append({"frame": n, "kind": "S1" if n % 2 == 0 else "S2", "bytes": len(pk), "md5": hashlib.md5(pk).hexdigest()})
json.dump({"config": "C1 320x240 yuv444p 8-bit x30, qp=0", "producer": "oracle/libffv2_oracle.so", "packets": out},
          open(os.path.join(ROOT, "tests", "golden", "c1_packets.json"), "w"), indent=1)
print(sum(p["bytes"] for p in out), "bytes in 30 packets")
