"""GPU tests of send_frame / receive_packet at qp > 0 on top of the lane coder (ffv2amd_qpring_*, round 3):
host frames in, host packets out, in send order, equal to the oracle's.  Parity unpinned (qp > 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ffmpeg_ffv2_amd import frames as synth  # noqa: E402


def _enc(w, h, fmt, max_batch=4):
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    return FFV2Encoder(w, h, fmt, device=0, max_batch=max_batch)


def _want(oracle, frame, fmt, qp):
    try:
        return oracle.encode(frame, fmt, qp=qp)
    except Exception:
        return None                                   # the reference would av_assert0


def _drain(enc, got, wait):
    from ffmpeg_ffv2_amd._lib import FFV2Error
    while True:
        try:
            r = enc.qpring_receive(wait=wait)
        except FFV2Error as e:
            assert e.code == -1, e.code
            got.append((e.tag, None))
            continue
        if r is None:
            return
        got.append(r)


@pytest.mark.parametrize("fmt,P,H,W,depth,qp", [("yuv444p", 3, 100, 150, 8, 16), ("gray", 1, 130, 70, 8, 4),
                                                ("yuv444p10le", 3, 64, 200, 10, 40)])
@pytest.mark.parametrize("per_call", [1, 3, 8])
def test_qpring_packets_in_order_equal_the_oracle(oracle, fmt, P, H, W, depth, qp, per_call):
    """13 frames (a structured one that usually aborts among them) through batches of 1 / 3 / 8: full batches leave on
    their own, the last partly filled one with the flush; EAGAIN from send is honoured by receiving; tags and packets
    come back in send order."""
    enc = _enc(W, H, fmt)
    n = 13
    frames = [synth.noise(11 * qp + i, P, H, W, depth) for i in range(n)]
    frames[5] = synth.make("S1", 5, P, H, W, depth)
    wide = np.zeros((P, H, W + 7), frames[7].dtype)                                 # rows further apart than the picture is wide
    wide[:, :, :W] = frames[7]
    frames[7] = wide[:, :, :W]
    enc.qpring_open(qp, per_call)
    got = []
    for i, f in enumerate(frames):
        while not enc.qpring_send(f, tag=100 + i):
            _drain(enc, got, wait=True)
        _drain(enc, got, wait=False)
    while not enc.qpring_flush():
        _drain(enc, got, wait=True)
    while enc.qpring_pending():
        _drain(enc, got, wait=True)
    assert [t for t, _ in got] == [100 + i for i in range(n)]
    for i, (t, pk) in enumerate(got):
        assert pk == _want(oracle, frames[i], fmt, qp), i
    assert enc.qpring_receive(wait=True) is None
    enc.qpring_close()
    enc.close()


@pytest.mark.parametrize("calls,backs", [(2, 1), (3, 2), (4, 4)])
def test_qpring_calls_in_flight_and_chains_side_by_side(oracle, monkeypatch, calls, backs):
    """The ring picks four calls in flight with a range chain each for small batches; FFV2AMD_QPRING_CALLS /
    FFV2AMD_LC_BACKS say otherwise.  Whatever they say: 23 frames in batches of 2 come back in order, equal to the
    oracle's, and `calls` batches in flight, one finished and waiting to be received and one full are accepted
    before a receive is due."""
    monkeypatch.setenv("FFV2AMD_QPRING_CALLS", str(calls))
    monkeypatch.setenv("FFV2AMD_LC_BACKS", str(backs))
    fmt, P, H, W, depth, qp = "yuv444p", 3, 90, 140, 8, 16
    enc = _enc(W, H, fmt)
    n = 23
    frames = [synth.noise(500 + i, P, H, W, depth) for i in range(n)]
    enc.qpring_open(qp, 2)
    got, sent = [], 0
    while sent < n and enc.qpring_send(frames[sent], tag=sent):
        sent += 1
    assert sent == 2 * (calls + 2)                    # the next send needs a place: EAGAIN until a batch has been received
    while sent < n:
        _drain(enc, got, wait=True)
        while sent < n and enc.qpring_send(frames[sent], tag=sent):
            sent += 1
    while not enc.qpring_flush():
        _drain(enc, got, wait=True)
    while enc.qpring_pending():
        _drain(enc, got, wait=True)
    assert [t for t, _ in got] == list(range(n))
    for i, (t, pk) in enumerate(got):
        assert pk == _want(oracle, frames[i], fmt, qp), i
    enc.qpring_close()
    enc.close()


def test_qpring_yuv420_pinned_and_phantom_w(oracle):
    """4:2:0 frames (up-converted on the ring's copy stream) and 4:4:4 frames with the phantom coefficient W, page-locked
    and pageable, mixed in one ring."""
    W_, H_, fmt, qp = 96, 80, "yuv444p", 16
    enc = _enc(W_, H_, fmt)
    rng = np.random.default_rng(5)
    f444 = [synth.noise(50 + i, 3, H_, W_, 8) for i in range(4)]
    cw, ch = (W_ + 1) // 2, (H_ + 1) // 2
    f420 = [(synth.noise(60 + i, 1, H_, W_, 8)[0], synth.noise(70 + i, 1, ch, cw, 8)[0], synth.noise(80 + i, 1, ch, cw, 8)[0])
            for i in range(3)]
    wv = rng.integers(-40, 40, (4, enc.info.block_planes)).astype(np.int32)
    pinned = enc.pinned_frames(2)
    pinned[:] = np.stack(f444[:2])
    enc.qpring_open(qp, 4)
    order = []
    assert enc.qpring_send(pinned[0], tag=0, pinned=True); order.append(("444", 0, None))
    assert enc.qpring_send(f420[0], tag=1, yuv420=True); order.append(("420", 0, None))
    assert enc.qpring_send(f444[2], tag=2, W=wv[2]); order.append(("444", 2, wv[2]))
    assert enc.qpring_send(pinned[1], tag=3, pinned=True, W=wv[1]); order.append(("444", 1, wv[1]))     # batch of 4 leaves
    assert enc.qpring_send(f420[1], tag=4, yuv420=True); order.append(("420", 1, None))
    assert enc.qpring_send(f444[3], tag=5); order.append(("444", 3, None))
    assert enc.qpring_send(f420[2], tag=6, yuv420=True); order.append(("420", 2, None))
    assert enc.qpring_flush()
    got = []
    while enc.qpring_pending():
        _drain(enc, got, wait=True)
    assert [t for t, _ in got] == list(range(7))
    for (kind, k, w), (_, pk) in zip(order, got):
        if kind == "444":
            try:
                want = oracle.encode(f444[k], fmt, qp=qp, W=w)
            except Exception:
                want = None
        else:
            y, u, v = f420[k]
            try:
                want = oracle.encode(oracle.sws_420_to_444(y, u, v, 8), fmt, qp=qp)
            except Exception:
                want = None
        assert pk == want, (kind, k)
    enc.qpring_close()
    enc.free_pinned()
    enc.close()


def test_qpring_arguments_and_reopen(oracle):
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = _enc(64, 64, "gray")
    with pytest.raises(FFV2Error) as e:
        enc.qpring_open(0, 4)
    assert e.value.code == -38
    with pytest.raises(FFV2Error):
        enc.qpring_open(16, 0)
    enc.qpring_open(16, 2)
    with pytest.raises(FFV2Error):
        enc.qpring_open(16, 2)                           # already open
    assert enc.qpring_receive(wait=True) is None         # nothing sent
    f = synth.noise(1, 1, 64, 64, 8)
    assert enc.qpring_send(f, tag=9)
    assert enc.qpring_receive(wait=True) is None         # not submitted yet: flush first
    assert enc.qpring_flush()
    assert enc.qpring_receive(wait=True) == (9, _want(oracle, f, "gray", 16))
    enc.qpring_close()
    enc.qpring_open(4, 3)                                # another qp, another batch size
    assert enc.qpring_send(f, tag=1) and enc.qpring_flush()
    assert enc.qpring_receive(wait=True) == (1, _want(oracle, f, "gray", 4))
    # the synchronous entry points still work beside a closed ring
    enc.qpring_close()
    assert enc.encode_batch_to_host(enc.upload(f[None]), qp=4) == [_want(oracle, f, "gray", 4)]
    enc.close()


def test_qpring_registers_pooled_pageable_frames(oracle):
    """FFV2AMD_FRAME_REGISTER: a pool of two ordinary buffers, refilled after their packets have come back; the ring
    page-locks each on first sight and reads it in place from then on."""
    W_, H_, fmt, qp = 960, 540, "yuv444p", 16              # planes of 506 KB: above the 256 KB below which nothing is page-locked
    enc = _enc(W_, H_, fmt)
    pool = [np.empty((3, H_, W_), np.uint8) for _ in range(2)]
    frames = [synth.noise(300 + i, 3, H_, W_, 8) for i in range(6)]
    enc.qpring_open(qp, 2)
    got = []
    for base in range(0, 6, 2):
        for k in range(2):
            pool[k][:] = frames[base + k]
            assert enc.qpring_send(pool[k], tag=base + k, register=True)
        # the batch of two is on its way; its packets back before the buffers are written again
        while len(got) < base + 2:
            _drain(enc, got, wait=True)
    assert [t for t, _ in got] == list(range(6))
    for i, (_, pk) in enumerate(got):
        assert pk == _want(oracle, frames[i], fmt, qp), i
    enc.qpring_close()
    enc.close()
