#!/usr/bin/env python3
"""GPU box: random-geometry soak of both T-stage kernels against the CPU oracle (bit-exact
coefficients, energies and qp = 0 packets; qp > 0 and 4:2:0 cases on the way, the latter through
the up-conversion kernel and through the frame ring).
usage: python tools/soak_parity.py [seconds] [seed]
tests/test_soak_gpu.py runs a fixed-seed slice of it (run(cases=...)) in the -m gpu suite."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FMTS = [("gray", 1, 8), ("yuv444p", 3, 8), ("gbrp", 3, 8), ("yuv444p10le", 3, 10), ("gbrp10le", 3, 10),
        ("yuv444p12le", 3, 12), ("gbrp12le", 3, 12)]


def run(budget=120.0, seed=2024, cases=None, max_w=900, max_h=700, p_qp=0.15, p_420=0.2, quiet=False):
    """Until `budget` seconds are over or `cases` geometries are done.  Returns (geometries, with qp > 0, with 4:2:0)."""
    from ffmpeg_ffv2_amd import FFV2Encoder, _lib, frames as synth, build
    from tests import oracle_lib
    build.build()
    oracle = oracle_lib.load()
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    t0, n, nq, n420 = time.time(), 0, 0, 0
    try:
        while (cases is None or n < cases) and (cases is not None or time.time() - t0 < budget):
            fmt, P, depth = FMTS[int(rng.integers(len(FMTS)))]
            pick = rng.random()
            W = int(rng.integers(1, 64)) if pick < 0.15 else int(rng.integers(1, max_w))
            H = int(rng.integers(1, 64)) if rng.random() < 0.15 else int(rng.integers(1, max_h))
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
            if rng.random() < p_qp and W * H < 200000:
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
            if fmt.startswith("yuv444p") and rng.random() < p_420 and W >= 8 and H >= 8:
                dt = synth.dtype_for(depth)
                y = rng.integers(0, 1 << depth, (H, W)).astype(dt)
                u = rng.integers(0, 1 << depth, ((H + 1) // 2, (W + 1) // 2)).astype(dt)
                v = rng.integers(0, 1 << depth, ((H + 1) // 2, (W + 1) // 2)).astype(dt)
                want = oracle.sws_420_to_444(y, u, v, depth)
                assert np.array_equal(enc.upconvert_420(y, u, v), want), ("420", depth, W, H)
                if n420 % 2 == 0:                       # and through the asynchronous ring, pageable planes
                    enc.ring_open(2)
                    assert enc.ring_send_420(y, u, v, tag=5)
                    tag, pk = enc.ring_receive()
                    assert tag == 5 and pk == oracle.encode(want, fmt), ("ring 420", depth, W, H, mode)
                    enc.ring_close()
                n420 += 1
            enc.close()
            n += 1
            if n % 500 == 0 and not quiet:
                print("  ... %d geometries, %.0f s" % (n, time.time() - t0), flush=True)
    finally:
        lib.ffv2amd_debug_force_tstage(-1)
    return n, nq, n420


if __name__ == "__main__":
    t0 = time.time()
    n, nq, n420 = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    print("soak ok: %d geometries (%d with qp > 0, %d with 4:2:0) in %.0f s" % (n, nq, n420, time.time() - t0))
