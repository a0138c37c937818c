#!/usr/bin/env python3
"""GPU box: random-geometry soak of both T-stage kernels against the CPU oracle (bit-exact
coefficients, energies and qp = 0 packets; a few qp > 0 and 4:2:0 cases on the way).
usage: python tools/soak_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffmpeg_ffv2_amd import FFV2Encoder, _lib, frames as synth, build  # noqa: E402
from tests import oracle_lib  # noqa: E402

build.build()
oracle = oracle_lib.load()
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
FMTS = [("gray", 1, 8), ("yuv444p", 3, 8), ("gbrp", 3, 8), ("yuv444p10le", 3, 10), ("gbrp10le", 3, 10),
        ("yuv444p12le", 3, 12), ("gbrp12le", 3, 12)]
t0, n, nq, n420 = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    fmt, P, depth = FMTS[int(rng.integers(len(FMTS)))]
    pick = rng.random()
    W = int(rng.integers(1, 64)) if pick < 0.15 else int(rng.integers(1, 900))
    H = int(rng.integers(1, 64)) if rng.random() < 0.15 else int(rng.integers(1, 700))
    F = int(rng.integers(1, 4))
    mode = int(rng.integers(0, 2))
    lib.ffv2amd_debug_force_tstage(mode)
    enc = FFV2Encoder(W, H, fmt, device=0, max_batch=F)
    kinds = ["S1", "S2", "flat"]
    frames = []
    for k in range(F):
        kind = kinds[int(rng.integers(3))]
        if kind == "flat":
            frames.append(np.full((P, H, W), int(rng.integers(1 << depth)), synth.dtype_for(depth)))
        else:
            frames.append(synth.make(kind, int(rng.integers(1 << 20)), P, H, W, depth))
    frames = np.stack(frames)
    dev = enc.upload(frames)
    coef, en = enc.tstage(dev)
    coef, en = coef.cpu().numpy(), en.cpu().numpy()
    got = enc.collect(*enc.encode_batch_device(dev))
    for k in range(F):
        co, eo = oracle.tstage(frames[k], fmt)
        assert np.array_equal(coef[k], co) and np.array_equal(en[k], eo), ("tstage", fmt, W, H, F, mode, k)
        assert got[k] == oracle.encode(frames[k], fmt), ("packet", fmt, W, H, F, mode, k)
    if rng.random() < 0.15 and W * H < 200000:
        qp = int(rng.choice([4, 16, 64]))
        noise = np.stack([synth.noise(int(rng.integers(1 << 20)), P, H, W, depth) for _ in range(F)])
        enc.set_device_coder(bool(rng.random() < 0.3) and W * H < 40000)     # sometimes the device range coder
        try:
            pk = enc.encode_batch_to_host(enc.upload(noise), qp=qp)
            for k in range(F):
                assert pk[k] == oracle.encode(noise[k], fmt, qp=qp), ("qp", qp, fmt, W, H, mode, k)
        except _lib.FFV2Error as ex:            # the reference would abort: the oracle must say so too
            assert ex.code == -1
            bad = False
            for k in range(F):
                try:
                    oracle.encode(noise[k], fmt, qp=qp)
                except Exception:
                    bad = True
            assert bad, ("abort only on the GPU side", qp, fmt, W, H)
        nq += 1
    if fmt.startswith("yuv444p") and rng.random() < 0.2 and W >= 8 and H >= 8:
        dt = synth.dtype_for(depth)
        y = rng.integers(0, 1 << depth, (H, W)).astype(dt)
        u = rng.integers(0, 1 << depth, ((H + 1) // 2, (W + 1) // 2)).astype(dt)
        v = rng.integers(0, 1 << depth, ((H + 1) // 2, (W + 1) // 2)).astype(dt)
        assert np.array_equal(enc.upconvert_420(y, u, v), oracle.sws_420_to_444(y, u, v, depth)), ("420", depth, W, H)
        n420 += 1
    enc.close()
    n += 1
    if n % 500 == 0:
        print("  ... %d geometries, %.0f s" % (n, time.time() - t0), flush=True)
lib.ffv2amd_debug_force_tstage(-1)
print("soak ok: %d geometries (%d with qp > 0, %d with 4:2:0) in %.0f s" % (n, nq, n420, time.time() - t0))
