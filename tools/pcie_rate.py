#!/usr/bin/env python3
"""Host-boundary (PCIe-inclusive) rate of ffv2amd_encode_frame: host frame in, host packet out,
one frame at a time, synchronous.  Reported in DESIGN.md, never the headline value."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
W, H, fmt, depth = 3840, 2160, "yuv444p10le", 10
enc = FFV2Encoder(W, H, fmt, device=0, max_batch=1)
fr = [synth.make("S1" if n % 2 == 0 else "S2", n, 3, H, W, depth) for n in range(4)]
for f in fr: enc.encode2(f)
t0 = time.perf_counter(); N = 20
for i in range(N): enc.encode2(fr[i % 4])
dt = (time.perf_counter() - t0) / N
print("encode2 host->host 4K yuv444p10le: %.3f ms/frame = %.1f Mpix/s (PCIe + pageable copies included)" % (dt * 1e3, W * H / dt / 1e6))
