#!/usr/bin/env python3
"""GPU box: random-geometry soak of both T-stage kernels against the CPU oracle (bit-exact
coefficients, energies and qp = 0 packets; qp > 0 and 4:2:0 cases on the way, the latter through
the up-conversion kernel and through the frame ring; frames with samples above their depth through the
wide path; packets through the decoder-side check against the oracle's decoder).
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
    trace = os.environ.get("FFV2_SOAK_TRACE")
    note = None                                        # the abort tracer (tools/debug/abort_trace.c) prints it if the process dies
    try:
        import ctypes
        so = os.path.join(ROOT, "tools", "debug", "abort_trace.so")
        if os.path.exists(so):
            note = (ctypes.c_char * 256).in_dll(ctypes.CDLL(so), "abort_trace_note")
    except (OSError, ValueError):
        note = None
    nwide = ndec = 0
    try:
        while (cases is None or n < cases) and (cases is not None or time.time() - t0 < budget):
            fmt, P, depth = FMTS[int(rng.integers(len(FMTS)))]
            pick = rng.random()
            W = int(rng.integers(1, 64)) if pick < 0.15 else int(rng.integers(1, max_w))
            H = int(rng.integers(1, 64)) if rng.random() < 0.15 else int(rng.integers(1, max_h))
            F = int(rng.integers(1, 4))
            mode = int(rng.integers(0, 2))
            lib.ffv2amd_debug_force_tstage(mode)
            if note is not None:
                note.value = ("soak_parity case %d: %s %dx%d, %d frames, forced T-stage kernel %d" % (n, fmt, W, H, F, mode)).encode()[:255]
            if trace:                                  # survives a process that dies: the case it died in
                with open(trace, "w") as tf:
                    tf.write("case %d: %s %dx%d, %d frames, forced T-stage kernel %d\n" % (n, fmt, W, H, F, mode))
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
            if depth > 8 and rng.random() < 0.08:
                # samples above the declared depth: the wide (plain int32) rerun + host-assembled packet (ffv2_wide.hip)
                wild = frames[0].copy()
                idx = rng.integers(0, wild.size, 1 + int(rng.integers(0, 6)))
                wild.reshape(-1)[idx] = rng.integers(1 << depth, 65536, idx.size)
                assert enc.encode2(wild) == oracle.encode(wild, fmt), ("wide", fmt, W, H, mode)
                nwide += 1
            if rng.random() < 0.1:
                # decoder-side check of the packet just made (ffv2amd_decode_frame) against the oracle's decoder
                pic, q0 = enc.decode(got[0], grid=bool(rng.integers(2)))
                assert q0 == 0
                ndec += 1
                want_pic, _ = oracle.decode(got[0], fmt, H, W, grid=False)
                on = (np.mgrid[0:H, 0:W][1] % 64 == 0) | (np.mgrid[0:H, 0:W][0] % 64 == 0)
                assert np.array_equal(pic[:, ~on], want_pic[:, ~on]), ("decode", fmt, W, H)
            if rng.random() < p_qp and W * H < 200000:
                qp = int(rng.choice([4, 16, 64]))
                noise = np.stack([synth.noise(int(rng.integers(1 << 20)), P, H, W, depth) for _ in range(F)])
                enc.set_device_coder(bool(rng.random() < 0.3) and W * H < 40000)     # sometimes the device range coder
                try:
                    pk = enc.encode_batch_to_host(enc.upload(noise), qp=qp)
                    for k in range(F):
                        assert pk[k] == oracle.encode(noise[k], fmt, qp=qp), ("qp", qp, fmt, W, H, mode, k)
                    if rng.random() < 0.3:
                        pic, q0 = enc.decode(pk[0])
                        want_pic, _ = oracle.decode(pk[0], fmt, H, W)
                        assert q0 == qp and np.array_equal(pic, want_pic), ("decode qp", qp, fmt, W, H)
                        ndec += 1
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
    if not quiet:
        print("  (%d frames above their depth through the wide path, %d packets through the decoder-side check)" % (nwide, ndec))
    return n, nq, n420


if __name__ == "__main__":
    t0 = time.time()
    n, nq, n420 = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    print("soak ok: %d geometries (%d with qp > 0, %d with 4:2:0) in %.0f s" % (n, nq, n420, time.time() - t0))
