#!/usr/bin/env python3
"""GPU box: random sizes / formats / qp / content through the many-frames-in-flight coder
(ffv2amd_lanecoder_*) against the host coder (ffv2amd_encode_batch_to_host), frame by frame.
Frames of very different lengths share a group of lanes on purpose ("quiet", "half"): the chain
kernel's LDS ring between its two wavefronts is ordered by hand (ffv2_lanecoder.hip), uneven
progress of the lanes is what would expose a mistake there.
usage: python tools/soak_lanecoder.py [cases=300] [seed=1]
tests/test_soak_gpu.py runs a fixed-seed slice of it in the -m gpu suite."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FMTS = [("gray", 1, 8), ("yuv444p", 3, 8), ("yuv444p10le", 3, 10), ("gbrp12le", 3, 12), ("yuv444p12le", 3, 12), ("gbrp", 3, 8)]


def run(cases=300, seed=1, max_w=400, max_h=300, max_frames=9, quiet=False, windows=(0, 0, 16, 48, 400, 4096, 65536)):
    """Returns (frames identical, aborts agreed, packet bytes compared).  windows: the coder's window sizes
    (symbols of the coding order cdf and chain work on at a time; 0 = default) drawn per case -- tiny ones cut
    the rows' chunks of 64 symbols at every possible place."""
    from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth, _lib
    from ffmpeg_ffv2_amd._lib import FFV2Error
    lib = _lib.load()
    rnd = random.Random(seed)
    checked = aborted = nbytes = 0
    t0 = time.time()
    for it in range(cases):
        fmt, P, depth = rnd.choice(FMTS)
        W, H = rnd.randint(1, max_w), rnd.randint(1, max_h)
        qp = rnd.randint(2, 64)
        n = rnd.randint(1, max_frames)
        enc = FFV2Encoder(W, H, fmt, device=0, max_batch=rnd.randint(1, 4))
        frames = []
        for i in range(n):
            k = rnd.choice(["S2", "S2", "S1", "quiet", "half"])
            if k == "quiet":
                f = synth.noise(it * 16 + i, P, H, W, depth)
                f = (f % rnd.choice([2, 5, 16]) + (1 << (depth - 1))).astype(f.dtype)
            elif k == "half":
                f = synth.noise(it * 16 + i, P, H, W, depth)
                f[:, H // 2:, :] = 1 << (depth - 1)
            else:
                f = synth.make(k, it * 16 + i, P, H, W, depth)
            frames.append(f)
        frames = np.stack(frames)
        dev = enc.upload(frames)
        lib.ffv2amd_debug_lanecoder_window(rnd.choice(windows))
        enc.lanecoder_open(n, rnd.choice([0, 0, 4096 + enc.info.block_planes * 600]))
        pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
        for i in range(n):
            try:
                want = enc.encode_batch_to_host(dev[i:i + 1], qp=qp)[0]
            except FFV2Error as e:
                assert status[i] == e.code, (it, i, status[i], e.code)
                aborted += 1
                continue
            if status[i] == -28:                      # tight packet_cap: allowed, must really not fit
                assert len(want) > 4096 + enc.info.block_planes * 600 - 64, (it, i, len(want))
                continue
            assert status[i] == 0 and pk[i, : sizes[i]].tobytes() == want, (it, i, fmt, W, H, qp)
            checked += 1
            nbytes += len(want)
        if it % 3 == 0:
            # the same frames from host memory through the qp > 0 ring (ffv2amd_qpring_*): batches of a random size,
            # packets in send order, equal to what the lane coder gave for the device-resident frames
            enc.lanecoder_close()
            enc.qpring_open(qp, rnd.randint(1, n + 1))
            got, sent, flushed = [], 0, False
            while len(got) < n:
                while sent < n and enc.qpring_send(frames[sent], tag=sent):
                    sent += 1
                if sent == n and not flushed:
                    flushed = enc.qpring_flush()
                try:
                    r = enc.qpring_receive(wait=True)
                except FFV2Error as e:
                    got.append((e.tag, e.code, None))
                    continue
                if r is not None:
                    got.append((r[0], 0, r[1]))
            for i, (tag, code, pkt) in enumerate(got):
                assert tag == i, (it, i, tag)
                if status[i] == -28:                  # the lane coder call above ran with a tight packet_cap; the ring does not
                    continue
                assert code == status[i], (it, i, code, status[i])
                if code == 0:
                    assert pkt == pk[i, : sizes[i]].tobytes(), (it, i, "qpring")
            enc.qpring_close()
        if it % 3 == 1 and n >= 2:
            # several calls in flight, a range chain each or shared (ffv2amd_lanecoder_open_ex): the frames in runs of
            # random lengths, every run a call; packets equal the single call's above
            calls = rnd.randint(2, 4)
            backs = rnd.randint(1, calls)
            enc.lanecoder_open(n, 0, calls_in_flight=calls, backs=backs)
            cuts = sorted(rnd.sample(range(1, n), min(n - 1, rnd.randint(1, 7))))
            runs = [(a, b) for a, b in zip([0] + cuts, cuts + [n])]
            sub = fin = 0
            while fin < len(runs):
                while sub < len(runs) and sub - fin < calls:
                    assert enc.lanecoder_submit(dev[runs[sub][0]:runs[sub][1]], qp)
                    sub += 1
                pk2, sizes2, status2 = enc.lanecoder_finish()
                a, b = runs[fin]
                for i in range(a, b):
                    if status[i] == -28:
                        continue
                    assert status2[i - a] == status[i], (it, i, "calls in flight", calls, backs)
                    if status[i] == 0:
                        assert pk2[i - a, : sizes2[i - a]].tobytes() == pk[i, : sizes[i]].tobytes(), (it, i, "calls in flight", calls, backs)
                fin += 1
        enc.close()
        lib.ffv2amd_debug_lanecoder_window(0)
        if it % 25 == 24 and not quiet:
            print("%d cases, %d frames identical, %d aborts agreed, %.0f s" % (it + 1, checked, aborted, time.time() - t0), flush=True)
    return checked, aborted, nbytes


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    checked, aborted, nbytes = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("soak_lanecoder: %d cases, %d frames identical (%.1f MB of packets), %d aborts agreed" % (cases, checked, nbytes / 1e6, aborted))
