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
    W, H, fmt, P, depth = 192, 128, "yuv444p10le", 3, 10
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = _enc(W, H, fmt)
    enc.ring_open(2)
    good = synth.make("S1", 0, P, H, W, depth)
    bad = good.copy()
    bad[1, 5, 7] = 1 << depth                     # above the declared depth
    assert enc.ring_send(bad, tag=1) and enc.ring_send(good, tag=2)
    with pytest.raises(FFV2Error) as ei:
        enc.ring_receive()
    assert ei.value.code == -34
    tag, pk = enc.ring_receive()
    assert tag == 2 and pk == oracle.encode(good, fmt)
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
