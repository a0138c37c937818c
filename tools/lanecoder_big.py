#!/usr/bin/env python3
"""Throughput of the many-frames-in-flight qp > 0 coder (GPU box only).
usage: python tools/lanecoder_big.py [frames_in_flight ...]   (1080p 8-bit noise, qp 16;
       GEOM=3840x2160:yuv444p10le and QP=n in the environment select another picture / qp)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth  # noqa: E402

geom, fmt = os.environ.get("GEOM", "1920x1080:yuv444p").split(":")
W, H = (int(v) for v in geom.split("x"))
P, depth = (1 if fmt.startswith("gray") else 3), (10 if "10" in fmt else 12 if "12" in fmt else 8)
qp = int(os.environ.get("QP", "16"))
counts = [int(a) for a in sys.argv[1:]] or [64, 256]
enc = FFV2Encoder(W, H, fmt, device=0, max_batch=int(os.environ.get("BATCH", "16")))
NB = int(os.environ.get("DISTINCT", "16"))
base = np.stack([synth.noise(n, P, H, W, depth) for n in range(NB)])
dbase = enc.upload(base)
host = enc.encode_batch_to_host(dbase, qp=qp)
print("host coder: %d packets, %d bytes each (about)" % (len(host), len(host[0])), flush=True)
print("scratch per frame in flight: %.1f MB" % (enc.lanecoder_bytes_per_frame() / 1e6), flush=True)
for F in counts:
    dev = dbase.repeat((F + NB - 1) // NB, *([1] * (dbase.dim() - 1)))[:F].contiguous()
    enc.lanecoder_open(F)
    stride = int(len(host[0]) * 1.3) + 4096
    pk, sizes, status = enc.lanecoder_encode(dev, qp, packet_stride=stride, as_arrays=True)     # warm-up + check
    bad = [i for i in range(F) if status[i] != 0 or pk[i, : sizes[i]].tobytes() != host[i % NB]]
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        enc.lanecoder_encode(dev, qp, packet_stride=stride, as_arrays=True)
    dt = (time.perf_counter() - t0) / reps
    print("F = %5d: %.3f s per call, %.1f frames/s, %.1f Mpix/s, mismatches %d" %
          (F, dt, F / dt, F * W * H / dt / 1e6, len(bad)), flush=True)
    enc.lanecoder_close()
    del dev
    torch.cuda.empty_cache()
