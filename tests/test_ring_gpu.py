"""GPU tests of the asynchronous frame ring (ffv2amd_ring_*): host frames in, host packets out,
H2D || T/E-stage || D2H on separate streams, packets delivered in send order and byte-identical
to the one-frame-at-a-time encode2 path and the CPU oracle."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ffmpeg_ffv2_amd import frames as synth  # noqa: E402


def _enc(w, h, fmt):
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    return FFV2Encoder(w, h, fmt, device=0, max_batch=1)


@pytest.mark.parametrize("fmt,P,H,W,depth", [("yuv444p", 3, 240, 320, 8), ("yuv444p10le", 3, 130, 200, 10),
                                             ("gray", 1, 65, 129, 8)])
@pytest.mark.parametrize("pinned", [False, True])
def test_ring_in_order_delivery_and_parity(oracle, fmt, P, H, W, depth, pinned):
    enc = _enc(W, H, fmt)
    depth_ring = 3
    enc.ring_open(depth_ring)
    frames = [synth.make("S2" if n % 2 else "S1", n, P, H, W, depth) for n in range(7)]
    want = [oracle.encode(f, fmt) for f in frames]
    if pinned:
        src = enc.pinned_frames(len(frames))
        for n, f in enumerate(frames):
            src[n] = f
    else:
        src = frames
    # the ring refuses a fourth frame in flight and keeps the three it has
    for n in range(depth_ring):
        assert enc.ring_send(src[n], tag=100 + n, pinned=pinned)
    assert enc.ring_pending() == depth_ring
    assert not enc.ring_send(src[3], tag=103, pinned=pinned)
    got = []
    sent = depth_ring
    rng = random.Random(5)
    while len(got) < len(frames):
        # random interleaving of sends and (non-)blocking receives
        if sent < len(frames) and rng.random() < 0.6 and enc.ring_send(src[sent], tag=100 + sent, pinned=pinned):
            sent += 1
            continue
        r = enc.ring_receive(wait=rng.random() < 0.5)
        if r is not None:
            got.append(r)
    assert enc.ring_receive() is None and enc.ring_pending() == 0      # EAGAIN on an empty ring
    assert [t for t, _ in got] == [100 + n for n in range(len(frames))], "delivery order"
    for n, (_, pk) in enumerate(got):
        assert pk == want[n], "packet %d differs from the oracle" % n
        assert pk == enc.encode2(frames[n])
    enc.ring_close()
    enc.free_pinned()
    enc.close()


def test_ring_later_frames_finish_while_the_oldest_is_held(oracle):
    """Completion runs ahead of delivery: with the whole ring in flight and nothing received, every
    frame finishes on the device; packets still come out oldest first."""
    import torch
    W, H, fmt, P, depth = 640, 360, "yuv444p10le", 3, 10
    enc = _enc(W, H, fmt)
    enc.ring_open(4)
    frames = [synth.make("S2", n, P, H, W, depth) for n in range(4)]
    for n, f in enumerate(frames):
        assert enc.ring_send(f, tag=n)
    torch.cuda.synchronize()                      # everything in flight has completed, none delivered
    for n in range(4):
        tag, pk = enc.ring_receive(wait=False)    # finished: the non-blocking form must deliver
        assert tag == n and pk == oracle.encode(frames[n], fmt)
    enc.ring_close()
    enc.close()


def test_ring_reports_a_bad_frame_and_moves_on(oracle):
    """A frame whose packet does not fit the caller's buffer fails alone (ENOSPC) and leaves the ring; a frame
    with a sample above the declared depth is no failure: the reference codes it, receive reruns it wide."""
    W, H, fmt, P, depth = 192, 128, "yuv444p10le", 3, 10
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = _enc(W, H, fmt)
    enc.ring_open(3)
    good = synth.make("S1", 0, P, H, W, depth)
    odd = good.copy()
    odd[1, 5, 7] = 1 << depth                     # above the declared depth
    assert enc.ring_send(good, tag=1) and enc.ring_send(odd, tag=2) and enc.ring_send(good, tag=3)
    full_out = enc._ring_out
    enc._ring_out = np.empty(16, np.uint8)        # too small for the first packet
    with pytest.raises(FFV2Error) as ei:
        enc.ring_receive()
    assert ei.value.code == -28
    enc._ring_out = full_out
    tag, pk = enc.ring_receive()
    assert tag == 2 and pk == oracle.encode(odd, fmt)
    tag, pk = enc.ring_receive()
    assert tag == 3 and pk == oracle.encode(good, fmt)
    enc.ring_close()
    enc.close()


def test_qp_pipeline_two_batches_in_flight(oracle):
    """ffv2amd_qp_submit / _finish: the GPU half of batch n+1 is issued before the host half of
    batch n runs; a third submit is refused; packets equal the synchronous path and the oracle
    (parity unpinned for qp > 0: the oracle restates the reference's PVQ asm)."""
    W, H, fmt, P, depth, qp = 200, 130, "yuv444p", 3, 8, 16
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    enc = FFV2Encoder(W, H, fmt, device=0, max_batch=3)
    batches = [np.stack([synth.noise(10 * b + n, P, H, W, depth) for n in range(3 - b)]) for b in range(3)]
    dev = [enc.upload(b) for b in batches]
    assert enc.qp_submit(dev[0], qp) and enc.qp_submit(dev[1], qp)
    assert not enc.qp_submit(dev[2], qp)              # two in flight already: EAGAIN
    got = [enc.qp_finish()]
    assert enc.qp_submit(dev[2], qp)
    got += [enc.qp_finish(), enc.qp_finish()]
    for b in range(3):
        assert len(got[b]) == 3 - b
        for n in range(3 - b):
            assert got[b][n] == oracle.encode(batches[b][n], fmt, qp=qp), (b, n)
    assert got[0] == enc.encode_batch_to_host(dev[0], qp=qp)
    enc.close()


@pytest.mark.parametrize("fmt,P,H,W,depth", [("gray", 1, 64, 64, 8), ("yuv444p", 3, 100, 150, 8), ("yuv444p10le", 3, 130, 200, 10)])
@pytest.mark.parametrize("qp", [4, 16, 64])
def test_device_range_coder_matches_host_coder_and_oracle(oracle, fmt, P, H, W, depth, qp):
    """SURVEY.md 8/A14 on the device: ffv2_rangecoder.hip (serial symbol loop on one lane per frame,
    carry propagation as a wavefront prefix scan) against the host coder and the oracle.
    Parity unpinned for qp > 0 (the oracle restates the reference's PVQ asm)."""
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    enc = FFV2Encoder(W, H, fmt, device=0, max_batch=3)
    frames = np.stack([synth.noise(7 * qp + n, P, H, W, depth) for n in range(3)])
    dev = enc.upload(frames)
    host = enc.encode_batch_to_host(dev, qp=qp)
    enc.set_device_coder(True)
    got = enc.encode_batch_to_host(dev, qp=qp)
    enc.set_device_coder(False)
    for n in range(3):
        assert got[n] == host[n] == oracle.encode(frames[n], fmt, qp=qp), n
    enc.close()


def test_device_range_coder_abort_conditions(oracle):
    """Where the reference would av_assert0 (daala_entropy.c:336: all pulses of a band on one
    coefficient; qp = 1) the device coder reports FFV2AMD_ERR_ABORT like the host coder."""
    from ffmpeg_ffv2_amd import FFV2Encoder
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = FFV2Encoder(64, 64, "gray", device=0, max_batch=1)
    enc.set_device_coder(True)
    flat = np.full((1, 1, 64, 64), 200, np.uint8)
    flat[0, 0, 10, 10] = 0                         # a single impulse: structured content concentrates pulses
    with pytest.raises(FFV2Error) as ei:
        enc.encode_batch_to_host(enc.upload(flat), qp=16)
    host_code = ei.value.code
    enc.set_device_coder(False)
    with pytest.raises(FFV2Error) as ei2:
        enc.encode_batch_to_host(enc.upload(flat), qp=16)
    assert host_code == ei2.value.code == -1
    enc.close()


# ---- 4:2:0 frames through the ring (SURVEY.md 8(f) rank 4 at the asynchronous boundary) ----
def _yuv420(seed, h, w, depth, kind="noise"):
    rng = np.random.default_rng(seed)
    dt = np.uint8 if depth == 8 else np.dtype("<u2")
    ch, cw = (h + 1) // 2, (w + 1) // 2
    if kind == "noise":
        return [rng.integers(0, 1 << depth, s).astype(dt) for s in ((h, w), (ch, cw), (ch, cw))]
    yy, xx = np.mgrid[0:ch, 0:cw]
    ramp = ((3 * xx + 5 * yy + seed) % (1 << depth)).astype(dt)
    y2, x2 = np.mgrid[0:h, 0:w]
    return [((7 * x2 + 3 * y2 + (x2 * y2 >> 6) + seed) % (1 << depth)).astype(dt), ramp, ramp[::-1].copy()]


FMT444 = {8: "yuv444p", 10: "yuv444p10le", 12: "yuv444p12le"}


@pytest.mark.parametrize("depth,h,w", [(8, 240, 320), (10, 130, 200), (12, 65, 129), (8, 37, 51), (10, 1080, 1920)])
@pytest.mark.parametrize("pinned", [False, True])
def test_ring_420_matches_convert_then_encode(oracle, depth, h, w, pinned):
    """yuv420p* frames through ffv2amd_ring_send_420: packets == oracle.encode(oracle.sws_420_to_444(frame)),
    delivered in send order, interleaved with 4:4:4 frames on the same ring.  Parity unpinned (swscale)."""
    fmt = FMT444[depth]
    enc = _enc(w, h, fmt)
    enc.ring_open(3)
    src = [_yuv420(20 + n, h, w, depth, "noise" if n % 2 else "ramp") for n in range(5)]
    conv = [oracle.sws_420_to_444(y, u, v, depth) for y, u, v in src]
    want = [oracle.encode(c, fmt) for c in conv]
    if pinned:
        pool = enc.pinned_frames_420(len(src))
        for (py, pu, pv), (y, u, v) in zip(pool, src):
            py[:] = y; pu[:] = u; pv[:] = v
        send = pool
    else:
        send = src
    order = [0, 1, 2, 99, 3, 4]              # 99: a 4:4:4 frame in between, on the same ring
    got, sent = [], 0
    while len(got) < len(order):
        while sent < len(order):
            t = order[sent]
            ok = enc.ring_send(conv[0], tag=99) if t == 99 else enc.ring_send_420(*send[t], tag=t, pinned=pinned)
            if not ok:
                break
            sent += 1
        got.append(enc.ring_receive(wait=True))
    tags = [t for t, _ in got]
    assert tags == [0, 1, 2, 99, 3, 4]
    for t, pk in got:
        assert pk == want[0 if t == 99 else t], "packet %d differs from convert-then-encode" % t
    enc.ring_close()
    enc.free_pinned()
    enc.close()


@pytest.mark.parametrize("depth,h,w", [(10, 2160, 3840), (12, 4320, 7680)])
def test_ring_420_literal_baseline_formats_full_size(oracle, depth, h, w):
    """BASELINE configs 3 and 5 as they are written (3840x2160 yuv420p10le, 7680x4320 yuv420p12le), host
    frame in, packet out; the up-converted picture is also held sample by sample to the oracle's."""
    fmt = FMT444[depth]
    enc = _enc(w, h, fmt)
    y, u, v = _yuv420(3, h, w, depth, "ramp")
    rng = np.random.default_rng(5)
    u[: h // 4] = rng.integers(0, 1 << depth, u[: h // 4].shape)       # noise and structure in one frame
    want444 = oracle.sws_420_to_444(y, u, v, depth)
    got444 = enc.upconvert_420(y, u, v)
    bad = np.argwhere(got444 != want444)
    assert len(bad) == 0, "first mismatch at (plane, y, x) = %s" % (bad[0],)
    enc.ring_open(2)
    assert enc.ring_send_420(y, u, v, tag=7)
    tag, pk = enc.ring_receive()
    assert tag == 7 and pk == oracle.encode(want444, fmt)
    enc.ring_close()
    enc.close()


def test_upconv_tiled_and_per_sample_kernels_agree(oracle, monkeypatch):
    """Odd geometries through both up-conversion kernels (FFV2AMD_UPCONV_NAIVE is read once per process, so
    the per-sample kernel is reached through a subprocess)."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\\n"
        "sys.path.insert(0, %r)\\n"
        "from ffmpeg_ffv2_amd import FFV2Encoder\\n"
        "from tests import oracle_lib\\n"
        "o = oracle_lib.load()\\n"
        "for depth, h, w in ((8, 67, 131), (10, 33, 257), (12, 200, 130), (8, 9, 300), (10, 17, 16)):\\n"
        "    rng = np.random.default_rng(h * w)\\n"
        "    dt = np.uint8 if depth == 8 else np.dtype('<u2')\\n"
        "    y, u, v = [rng.integers(0, 1 << depth, s).astype(dt) for s in ((h, w), ((h + 1) // 2, (w + 1) // 2), ((h + 1) // 2, (w + 1) // 2))]\\n"
        "    e = FFV2Encoder(w, h, {8: 'yuv444p', 10: 'yuv444p10le', 12: 'yuv444p12le'}[depth])\\n"
        "    assert (e.upconvert_420(y, u, v) == o.sws_420_to_444(y, u, v, depth)).all(), (depth, h, w)\\n"
        "    e.close()\\n"
        "print('ok')\\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for naive in ("0", "1"):
        env = dict(os.environ, FFV2AMD_UPCONV_NAIVE=naive)
        r = subprocess.run([sys.executable, "-c", code.replace("\\n", "\n")], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "ok" in r.stdout, (naive, r.stdout, r.stderr)


def test_ring_registers_pooled_pageable_frames(oracle):
    """FFV2AMD_FRAME_REGISTER: ordinary memory from a pool of long-lived buffers (what libavcodec's frame pools hand
    out) is page-locked by the ring the first time it sees a buffer and read in place afterwards.  Same packets,
    same order; new contents in a registered buffer are picked up; unflagged and 4:2:0 frames mix in."""
    W, H, fmt, P, depth = 640, 360, "yuv444p10le", 3, 10
    enc = _enc(W, H, fmt)
    enc.ring_open(3)
    pool = [np.zeros((P, H, W), "<u2") for _ in range(3)]              # the caller's frame pool
    frames = [synth.make("S2" if n % 2 else "S1", n, P, H, W, depth) for n in range(10)]
    want = [oracle.encode(f, fmt) for f in frames]
    got, sent, free = [], 0, [0, 1, 2]
    inflight = []
    while len(got) < len(frames):
        while sent < len(frames) and free:
            b = free.pop(0)
            pool[b][:] = frames[sent]                                  # the buffer is refilled only once its packet is out
            assert enc.ring_send(pool[b], tag=sent, register=True)
            inflight.append(b)
            sent += 1
        got.append(enc.ring_receive())
        free.append(inflight.pop(0))
    assert got == list(enumerate(want))
    # a 4:2:0 frame from ordinary memory with the flag, an unflagged frame and a page-locked one behind it
    rng = np.random.default_rng(3)
    y, u, v = [rng.integers(0, 1 << depth, s).astype("<u2") for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))]
    pin = enc.pinned_frames(1)
    pin[0] = frames[1]
    assert enc.ring_send_420(y, u, v, tag=50, register=True) and enc.ring_send(frames[0], tag=51) and enc.ring_send(pin[0], tag=52, pinned=True)
    assert enc.ring_receive() == (50, oracle.encode(oracle.sws_420_to_444(y, u, v, depth), fmt))
    assert enc.ring_receive() == (51, want[0]) and enc.ring_receive() == (52, want[1])
    enc.ring_close()                                                   # ends the registrations
    enc.free_pinned()
    enc.ring_open(2)                                                   # a new ring registers afresh
    assert enc.ring_send(pool[0], tag=9, register=True)
    assert enc.ring_receive()[0] == 9
    enc.ring_close()
    enc.close()
