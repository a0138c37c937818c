"""The AVCodec-shaped shim as a frame-level fan-out (SURVEY.md 8(e) behind the C boundary): ONE
context, a list of devices, frame n on device n % G, packets back in send order
(ffv2enc.c:461-469: frames are independent; encode.c:420,449: one thread sends and receives).
The one-GPU box rehearses it with the same ordinal twice -- two encoders, two rings."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ffmpeg_ffv2_amd import frames as synth  # noqa: E402
from tests.codec_ctypes import FRAME_YUV420, Packet, frame_of, make_ctx  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from ffmpeg_ffv2_amd import _lib, build
    build.build()
    return _lib.load()


def _drive(lib, ctx, frames, flags=0, first_pts=500):
    """send / receive until every frame has come back; returns [(pts, bytes or negative code)]."""
    out, sent = [], 0
    while len(out) < len(frames):
        while sent < len(frames):
            r = lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[sent], first_pts + sent)), flags)
            if r == -11:
                break
            assert r == 0, r
            sent += 1
        pkt = Packet()
        r = lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1)
        assert r != -11
        if r < 0:
            out.append((None, r))
            continue
        out.append((pkt.pts, bytes(pkt.data[: pkt.size])))
        lib.ffv2amd_packet_unref(C.byref(pkt))
    assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(Packet()), 1) == -11
    return out


@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
def test_device_list_delivers_in_send_order(oracle, devices):
    lib = _lib()
    W, H = 320, 240
    frames = [synth.make("S2" if n % 3 else "S1", n, 3, H, W, 8) for n in range(11)]
    ctx = make_ctx(W, H, 5, ring_depth=2, devices=devices)
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    got = _drive(lib, ctx, frames)
    assert [p for p, _ in got] == [500 + n for n in range(len(frames))]
    for n, (_, pk) in enumerate(got):
        assert pk == oracle.encode(frames[n], "yuv444p"), n
    # the list holds 2 (3) rings of depth 2: that many frames may be in flight before EAGAIN
    cap = 2 * (len(devices) if devices else 1)
    for n in range(cap):
        assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[n], n)), 0) == 0
    assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[0], 99)), 0) == -11
    for n in range(cap):
        pkt = Packet()
        assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1) == 0 and pkt.pts == n
        lib.ffv2amd_packet_unref(C.byref(pkt))
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0


def test_device_list_equals_single_ring_byte_for_byte(oracle):
    lib = _lib()
    W, H = 200, 130
    frames = [synth.make("S2", 40 + n, 3, H, W, 10) for n in range(9)]
    outs = []
    for devices in (None, [0, 0]):
        ctx = make_ctx(W, H, 70, ring_depth=3, devices=devices)
        assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
        outs.append(_drive(lib, ctx, frames))
        assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0
    assert outs[0] == outs[1]


def test_a_frame_above_its_depth_keeps_its_place(oracle):
    """ffv2.c:26-38 level-shifts any 16-bit sample: the frame is coded (through the wide path), in order."""
    lib = _lib()
    W, H = 192, 128
    frames = [synth.make("S1", n, 3, H, W, 10) for n in range(5)]
    frames[2] = frames[2].copy()
    frames[2][0, 3, 3] = 1 << 10
    ctx = make_ctx(W, H, 70, ring_depth=2, devices=[0, 0])
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    got = _drive(lib, ctx, frames)
    for n in range(5):
        assert got[n] == (500 + n, oracle.encode(frames[n], "yuv444p10le")), n
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0


def test_bad_device_list_is_refused():
    lib = _lib()
    ctx = make_ctx(64, 64, 5, devices=[0, 12345])
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == -5 and not ctx.priv_data
    ctx = make_ctx(64, 64, 5)
    ctx.nb_devices = 17
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == -22 and not ctx.priv_data


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_yuv420_frames_through_send_frame(oracle, devices):
    lib = _lib()
    W, H, depth = 322, 242, 10
    rng = np.random.default_rng(9)
    src = [[rng.integers(0, 1 << depth, s).astype("<u2") for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))] for _ in range(6)]
    ctx = make_ctx(W, H, 70, ring_depth=2, devices=devices)
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    got = _drive(lib, ctx, src, flags=FRAME_YUV420)
    for n, (pts, pk) in enumerate(got):
        assert pts == 500 + n
        assert pk == oracle.encode(oracle.sws_420_to_444(*src[n], depth), "yuv444p10le"), n
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_send_receive_at_qp_above_zero(oracle, devices):
    """global_quality > 0 through send_frame / receive_packet (ffv2amd_qp_send_frame): two frames in flight
    per device, packets == oracle at the same qp (parity unpinned for qp > 0), a frame the reference would
    abort on comes back as -1 and the next one follows."""
    lib = _lib()
    W, H, qp = 200, 130, 16
    frames = [synth.noise(70 + n, 3, H, W, 8) for n in range(6)]
    flat = np.full((3, H, W), 200, np.uint8)
    flat[0, 10, 10] = 0                                       # a lone impulse concentrates a band's pulses: av_assert0
    frames[3] = flat
    ctx = make_ctx(W, H, 5, qp=qp, devices=devices)
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    got = _drive(lib, ctx, frames)
    for n, (pts, pk) in enumerate(got):
        if n == 3:
            assert (pts, pk) == (None, -1)
        else:
            assert pts == 500 + n and pk == oracle.encode(frames[n], "yuv444p", qp=qp), n
    # in flight: 2 per device, then EAGAIN; global_quality may not change meanwhile
    cap = 2 * (len(devices) if devices else 1)
    good = [f for n, f in enumerate(frames) if n != 3]
    for n in range(cap):
        assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(good[n], n)), 0) == 0
    assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[0], 9)), 0) == -11
    ctx.global_quality = 0
    assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[0], 9)), 0) == -22
    ctx.global_quality = qp
    for n in range(cap):
        pkt = Packet()
        assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1) == 0 and pkt.pts == n
        lib.ffv2amd_packet_unref(C.byref(pkt))
    # with nothing in flight the mode may change: qp 0 through the ring of the same context
    ctx.global_quality = 0
    got0 = _drive(lib, ctx, frames[:2])
    assert [pk for _, pk in got0] == [oracle.encode(f, "yuv444p") for f in frames[:2]]
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no outer launcher must start two ranks itself (the flag
    used to be parsed and ignored).  gloo lets the two ranks share this box's single GPU; the numbers mean
    nothing, the control path is what is checked."""
    import json
    env = dict(os.environ, FFV2_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--config", "C2", "--frames-per-step", "2", "--host-frames", "6", "--no-preroll",
                        "--steady-seconds", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    hb = res["host_boundary"]
    assert hb["ranks_seen"] == 2 and hb["packets_gathered"] == 12 and hb["packets_match_device_path"]


def test_yuv420_frames_at_qp_above_zero(oracle):
    """4:2:0 sources at global_quality > 0: up-conversion + qp pipeline behind send_frame / receive_packet."""
    lib = _lib()
    W, H, depth, qp = 200, 130, 8, 16
    rng = np.random.default_rng(4)
    src = [[rng.integers(0, 1 << depth, s).astype(np.uint8) for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))] for _ in range(4)]
    ctx = make_ctx(W, H, 5, qp=qp, devices=[0, 0])
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    got = _drive(lib, ctx, src, flags=FRAME_YUV420)
    for n, (pts, pk) in enumerate(got):
        conv = oracle.sws_420_to_444(*src[n], depth)
        try:
            want = oracle.encode(conv, "yuv444p", qp=qp)
        except RuntimeError:
            want = None
        assert (pts, pk) == ((500 + n, want) if want is not None else (None, -1)), n
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_send_receive_at_qp_above_zero_through_the_device_coder(oracle, devices):
    """qp_frames_per_call > 0: frames are collected per device and coded side by side on the device
    (ffv2amd_qpring_*); send_frame(NULL) drains; order, pts and packets as with the one-frame pipeline; 4:2:0 frames
    take the same road."""
    lib = _lib()
    W, H, qp = 200, 130, 16
    frames = [synth.noise(170 + n, 3, H, W, 8) for n in range(11)]
    flat = np.full((3, H, W), 200, np.uint8)
    flat[0, 10, 10] = 0
    frames[4] = flat                                           # the reference would av_assert0
    ctx = make_ctx(W, H, 5, qp=qp, devices=devices, qp_frames_per_call=3)
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    out, sent, drained = [], 0, False
    while len(out) < len(frames):
        while sent < len(frames):
            r = lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(frames[sent], 900 + sent)), 0)
            if r == -11:
                break
            assert r == 0, r
            sent += 1
        if sent == len(frames) and not drained:
            r = lib.ffv2amd_codec_send_frame(C.byref(ctx), None, 0)                  # drain; EAGAIN: take packets first
            assert r in (0, -11), r
            drained = r == 0
        pkt = Packet()
        r = lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1)
        if r == -11:
            assert not drained
            continue
        if r < 0:
            out.append((None, r))
            continue
        out.append((pkt.pts, bytes(pkt.data[: pkt.size])))
        lib.ffv2amd_packet_unref(C.byref(pkt))
    for n, (pts, pk) in enumerate(out):
        if n == 4:
            assert (pts, pk) == (None, -1)
        else:
            assert pts == 900 + n and pk == oracle.encode(frames[n], "yuv444p", qp=qp), n
    # another qp with nothing in flight: the rings are reopened; 4:2:0 frames
    ctx.global_quality = 4
    cw, ch = (W + 1) // 2, (H + 1) // 2
    f420 = [(synth.noise(30 + n, 1, H, W, 8)[0], synth.noise(40 + n, 1, ch, cw, 8)[0], synth.noise(50 + n, 1, ch, cw, 8)[0])
            for n in range(4)]
    for n, f in enumerate(f420):
        assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(f, n)), FRAME_YUV420) == 0
    assert lib.ffv2amd_codec_send_frame(C.byref(ctx), None, 0) == 0
    for n, (y, u, v) in enumerate(f420):
        pkt = Packet()
        assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1) == 0 and pkt.pts == n
        assert bytes(pkt.data[: pkt.size]) == oracle.encode(oracle.sws_420_to_444(y, u, v, 8), "yuv444p", qp=4), n
        lib.ffv2amd_packet_unref(C.byref(pkt))
    assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(Packet()), 1) == -11
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0
