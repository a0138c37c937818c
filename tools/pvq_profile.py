#!/usr/bin/env python3
"""GPU box: ffv2_pvq_kernel alone on 16 1080p noise frames, 5 launches (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools/pvq_profile.py [qp=16]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth, _lib as L  # noqa: E402

qp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = L.load()
enc = FFV2Encoder(1920, 1080, "yuv444p", device=0, max_batch=16)
fr = np.stack([synth.noise(n, 3, 1080, 1920, 8) for n in range(16)])
d = enc.upload(fr)
ms = C.c_float(0)
L.check(lib.ffv2amd_debug_pvq_time(enc._h, 16, d.data_ptr(), qp, 5, C.byref(ms)), "pvq_time")
print("ffv2_pvq_kernel qp %d: %.3f ms per 16 frames (%d block-planes)" % (qp, ms.value, 16 * enc.info.block_planes))
enc.close()
