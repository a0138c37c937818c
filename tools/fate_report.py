#!/usr/bin/env python3
"""FATE-style encode -> decode report for FFV2 (the shape of enc_dec, reference tests/fate-run.sh:188-210,
whose output FATE diffs against tests/ref/vsynth/*): four lines per case --

    <md5 of the packet stream>  *<name>.ffv2
    <bytes> <name>.ffv2
    <md5 of the decoded raw video>  *<name>.out.rawvideo
    stddev: .. PSNR: .. MAXDIFF: .. bytes: <source>/ <decoded>      (tests/tiny_psnr.c, byte-wise as FATE's vsynth tests)

Cases = BASELINE configs C1-C3 as they reach encode2() (4:4:4), qp 0 and 16.  Sources are the synthetic
frames of SURVEY.md 8(d): structured / noise alternating at qp 0, noise only at qp 16 (structured
content makes the reference abort at qp > 0).  The decoder is the reference's as it is (ffv2dec.c), grid
overwrite of its `#define DEBUGGING` included; note that at qp 0 it divides by sqrt(0) (ffv2dec.c:134): the
decoded picture is garbage by construction and the PSNR line says so.  The reference's 8-bit text overlay
(decoding time) is not reproducible and left out.

  python tools/fate_report.py --oracle [--write]   CPU oracle (both directions); --write refreshes tests/golden/fate/*
  python tools/fate_report.py [cases...]           GPU: packets through the C-ABI (ffv2amd_encode_frame),
                                                   decode through ffv2amd_decode_frame; diffs against tests/golden/fate/*
PARITY UNPINNED: the reference holds no FFV2 FATE reference; the fixtures are this repo's oracle's output."""
import argparse
import hashlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden", "fate")

# name: (width, height, pix_fmt, planes, depth, frames)
CONFIGS = {
    "C1": (320, 240, "yuv444p", 3, 8, 30),
    "C2": (1920, 1080, "yuv444p", 3, 8, 4),
    "C3": (3840, 2160, "yuv444p10le", 3, 10, 2),
}
QPS = (0, 16)


def case_names():
    return ["ffv2-%s-qp%d" % (c, q) for c in CONFIGS for q in QPS]


def source_frames(cfg, qp):
    from ffmpeg_ffv2_amd import frames as synth
    W, H, fmt, P, depth, n = CONFIGS[cfg]
    return [synth.make("S1" if (k % 2 == 0 and qp == 0) else "S2", k, P, H, W, depth) for k in range(n)]


def tiny_psnr_line(src, dec):
    """tests/tiny_psnr.c run_psnr with len = 1 (bytes), its formulas in floating point."""
    a = np.frombuffer(src, np.uint8).astype(np.int64)
    b = np.frombuffer(dec, np.uint8).astype(np.int64)
    n = min(a.size, b.size)
    d = a[:n] - b[:n]
    sse = int((d * d).sum())
    dev = math.sqrt(sse / max(n, 1))
    psnr = 10 * math.log10(255.0 * 255.0 * n / sse) if sse else 999.99
    return "stddev:%8.2f PSNR:%6.2f MAXDIFF:%5d bytes:%9d/%9d" % (dev, psnr, int(np.abs(d).max()) if n else 0, a.size, b.size)


def report(name, packets, sources, decoded):
    stream = b"".join(packets)
    src = b"".join(f.tobytes() for f in sources)
    dec = b"".join(f.tobytes() for f in decoded)
    return "\n".join(["%s *%s.ffv2" % (hashlib.md5(stream).hexdigest(), name),
                      "%d %s.ffv2" % (len(stream), name),
                      "%s *%s.out.rawvideo" % (hashlib.md5(dec).hexdigest(), name),
                      tiny_psnr_line(src, dec)]) + "\n"


def run_oracle(name):
    from tests import oracle_lib
    o = oracle_lib.load()
    cfg, qp = name.split("-")[1], int(name.split("qp")[1])
    W, H, fmt, P, depth, n = CONFIGS[cfg]
    frames = source_frames(cfg, qp)
    packets = [o.encode(f, fmt, qp=qp) for f in frames]
    decoded = [o.decode(pk, fmt, H, W, grid=True)[0] for pk in packets]
    return report(name, packets, frames, decoded)


def run_gpu(name):
    from ffmpeg_ffv2_amd import FFV2Encoder
    cfg, qp = name.split("-")[1], int(name.split("qp")[1])
    W, H, fmt, P, depth, n = CONFIGS[cfg]
    frames = source_frames(cfg, qp)
    enc = FFV2Encoder(W, H, fmt, device=0, max_batch=1)
    packets = [enc.encode2(f, qp=qp) for f in frames]
    decoded = []
    for pk in packets:
        pic, q = enc.decode(pk, grid=True)
        assert q == qp
        decoded.append(pic)
    enc.close()
    return report(name, packets, frames, decoded)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", help="default: all of %s" % ", ".join(case_names()))
    ap.add_argument("--oracle", action="store_true", help="CPU oracle instead of the GPU path")
    ap.add_argument("--write", action="store_true", help="with --oracle: refresh tests/golden/fate/*")
    args = ap.parse_args()
    bad = 0
    for name in args.cases or case_names():
        text = run_oracle(name) if args.oracle else run_gpu(name)
        sys.stdout.write(text)
        ref = os.path.join(GOLDEN, name)
        if args.oracle and args.write:
            os.makedirs(GOLDEN, exist_ok=True)
            open(ref, "w").write(text)
        elif os.path.exists(ref) and open(ref).read() != text:
            print("--- %s differs from tests/golden/fate/%s" % (name, name))
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
