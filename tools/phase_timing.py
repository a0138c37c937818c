#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the T-stage kernel (development aid, GPU box only).

Builds a second library with -DFFV2_PHASE_TIMING under tools/phase/ (the shipped library is not
touched), runs the benchmark workload through it and prints the share of each phase.
usage: python tools/phase_timing.py build   (in the container: cross-compiles)
       python tools/phase_timing.py run     (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "phase")
SO = os.path.join(OUT, "libffv2amd_timing.so")
NAMES = ["entry+table loads", "A load/level shift", "B horizontal lapping", "C vertical lapping",
         "D column DCT+transpose", "E row DCT", "F gather/energy/gains", "coefficient stores"]


def build():
    from ffmpeg_ffv2_amd import build as b
    b.build()
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "ffv2_kernels_timing.o")
    subprocess.run([b._hipcc()] + b.HIPFLAGS + ["-DFFV2_PHASE_TIMING", "-c", os.path.join(b.CSRC, "ffv2_kernels.hip"),
                                              "-o", obj], check=True)
    objs = [obj] + [os.path.join(b.CSRC, n) for n in ("ffv2_pvq.o", "ffv2_inverse.o", "ffv2_capi.o", "ffv2enc_amd.o")]
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs, check=True)
    print(SO)


def run():
    import numpy as np
    import ffmpeg_ffv2_amd._lib as L
    L.SO = SO
    from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
    import torch
    W, H, fmt, nf = 3840, 2160, "yuv444p10le", 8
    enc = FFV2Encoder(W, H, fmt, max_batch=nf)
    fr = np.stack([synth.make("S1" if n % 2 == 0 else "S2", n, 3, H, W, 10) for n in range(nf)])
    dev = enc.upload(fr)
    coef = torch.empty((nf, enc.info.block_planes, 4096), dtype=torch.int32, device=dev.device)
    enc.set_coef_sink(coef)
    pk = enc.alloc_packets(nf)
    stream = torch.cuda.current_stream(dev.device).cuda_stream
    lib = L.load()
    lib.ffv2amd_debug_phase_ticks.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.ffv2amd_debug_phase_ticks.restype = None
    lib.ffv2amd_debug_phase_alloc.argtypes = [C.c_size_t]
    groups = (enc.info.block_planes + 7) // 8 * 8 * nf     # upper bound on workgroups
    assert lib.ffv2amd_debug_phase_alloc(groups) == 0
    ticks = (C.c_ulonglong * 8)()
    for _ in range(100):
        enc.encode_batch_device(dev, out=pk, stream=stream)
    torch.cuda.synchronize()
    lib.ffv2amd_debug_phase_ticks(ticks)          # per-workgroup ticks of the last launch
    t = np.array(list(ticks), dtype=np.float64)
    nblk = enc.info.block_planes * nf
    print("ticks per block-plane (one wave): %.0f" % (t.sum() / nblk))
    for n, v in zip(NAMES, t):
        print("  %-26s %8.0f  %5.1f %%" % (n, v / nblk, 100 * v / t.sum()))


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
