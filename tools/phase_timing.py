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
WALL = "--wall" in sys.argv      # build/run the variant that records wave start + life (slots 0 and 2)
NAMES = ["entry+table loads", "A load/level shift", "B horizontal lapping", "C vertical lapping",
         "D column DCT+transpose", "E row DCT", "F gather/energy/gains", "coefficient stores"]


def build():
    from ffmpeg_ffv2_amd import build as b
    b.build()
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "ffv2_kernels_timing.o")
    subprocess.run([b._hipcc()] + b.HIPFLAGS + ["-DFFV2_PHASE_TIMING"] + (["-DFFV2_PHASE_WALL"] if WALL else []) + ["-c", os.path.join(b.CSRC, "ffv2_kernels.hip"),
                                              "-o", obj], check=True)
    objs = [obj] + [os.path.join(b.CSRC, n) for n in ("ffv2_pvq.o", "ffv2_inverse.o", "ffv2_upconv.o", "ffv2_rangecoder.o", "ffv2_lanecoder.o", "ffv2_capi.o", "ffv2enc_amd.o", "ffv2mkv.o")]
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
    groups = (enc.info.block_planes + 7) // 8 * 8 * nf     # workgroups per launch
    assert lib.ffv2amd_debug_phase_alloc(groups) == 0
    ticks = (C.c_ulonglong * 8)()
    for _ in range(100):
        enc.encode_batch_device(dev, out=pk, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        enc.encode_batch_device(dev, out=pk, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    print("instrumented build: %.4f ms per call (T-stage + E-stage)" % (e0.elapsed_time(e1) / 50))
    lib.ffv2amd_debug_phase_ticks(ticks)          # per-workgroup ticks of the last launch
    if not WALL:
        t = np.array(list(ticks), dtype=np.float64)
        nblk = enc.info.block_planes * nf
        print("ticks per block-plane (one wave, 2.4 GHz s_memtime): %.0f" % (t.sum() / nblk))
        for n, v in zip(NAMES, t):
            print("  %-26s %8.0f  %5.1f %%" % (n, v / nblk, 100 * v / t.sum()))
        return
    raw = np.zeros((groups, 8), dtype=np.uint64)
    lib.ffv2amd_debug_phase_raw.argtypes = [C.c_void_p]
    lib.ffv2amd_debug_phase_raw(raw.ctypes.data)
    st = raw[:, 0].astype(np.int64)
    du = raw[:, 2].astype(np.int64)

    live = raw[:, 0] != 0
    st, du = st[live], du[live]
    span = (st + du).max() - st.min()
    print("waves %d, mean life %.2f us, launch span %.1f us -> mean waves in flight %.0f (%.1f per CU)" %
          (live.sum(), du.mean() / 100.0, span / 100.0, du.sum() / span, du.sum() / span / 256))
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([st + du, -np.ones_like(st)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    inflight = np.cumsum(ev[:, 1])
    print("peak waves in flight %d" % inflight.max())
    t0 = st.min()
    edges = np.linspace(0, span, 21)
    idx = np.searchsorted(ev[:, 0] - t0, edges[1:-1])
    print("in flight at 5% steps of the launch:", [int(inflight[min(i, len(inflight) - 1)]) for i in idx])
    print("life (us) percentiles 5/50/95/99/max: %s" % np.round(np.percentile(du, [5, 50, 95, 99, 100]) / 100.0, 1))


if __name__ == "__main__":
    (build if "build" in sys.argv[1:] else run)()
