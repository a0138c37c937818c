"""Frames the fast kernels refuse: samples above the declared bit depth, band gains beyond the device's
threshold table.  The reference has no such notion -- ref_2_coeffs_10/12 shift whatever 16-bit value they
read (ffv2.c:26-38) and quant_block codes any gain (ffv2enc.c:174) -- so every entry point that ends in
host memory at qp == 0 reruns such a frame through the plain-int32 T-stage (ffv2_wide.hip) and assembles
the packet on the host: same bytes as the oracle.  Paths with no host in the loop report ERANGE."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ffmpeg_ffv2_amd import frames as synth  # noqa: E402


def _enc(w, h, fmt, max_batch=1):
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    return FFV2Encoder(w, h, fmt, device=0, max_batch=max_batch)


def _wild(seed, P, H, W, kind):
    rng = np.random.default_rng(seed)
    if kind == "full16":                      # every sample anywhere in 16 bits
        return rng.integers(0, 65536, (P, H, W)).astype("<u2")
    if kind == "spikes":                      # picture content with a few wild samples
        f = synth.make("S1", seed, P, H, W, 10).astype("<u2")
        idx = rng.integers(0, f.size, 9)
        f.reshape(-1)[idx] = rng.integers(1024, 65536, 9)
        return f
    f = np.full((P, H, W), 65535, "<u2")      # the largest level everywhere
    f[:, ::7, ::5] = 0
    return f


@pytest.mark.parametrize("fmt,depth", [("yuv444p10le", 10), ("gbrp12le", 12)])
@pytest.mark.parametrize("H,W", [(64, 64), (130, 200), (65, 257)])
@pytest.mark.parametrize("kind", ["full16", "spikes", "max"])
def test_wide_tstage_matches_oracle(oracle, fmt, depth, H, W, kind):
    enc = _enc(W, H, fmt)
    fr = _wild(H * W + depth, 3, H, W, kind)
    coef, en = enc.tstage_wide(enc.upload(fr[None]))
    co, eo = oracle.tstage(fr, fmt)
    assert np.array_equal(coef.cpu().numpy(), co) and np.array_equal(en.cpu().numpy(), eo)
    # and on ordinary picture data it is the fast kernels' result
    ok = synth.make("S2", 3, 3, H, W, depth)
    d = enc.upload(ok[None])
    cw, ew = enc.tstage_wide(d)
    cf, ef = enc.tstage(d)
    assert np.array_equal(cw.cpu().numpy(), cf[0].cpu().numpy()) and np.array_equal(ew.cpu().numpy(), ef[0].cpu().numpy())
    enc.close()


@pytest.mark.parametrize("fmt,depth,kind", [("yuv444p10le", 10, "full16"), ("yuv444p12le", 12, "spikes"), ("gbrp10le", 10, "max")])
def test_every_host_ended_path_codes_the_frame_like_the_reference(oracle, fmt, depth, kind):
    H, W = 200, 330
    enc = _enc(W, H, fmt, max_batch=3)
    wild = _wild(5, 3, H, W, kind)
    good = synth.make("S2", 1, 3, H, W, depth)
    want_w, want_g = oracle.encode(wild, fmt), oracle.encode(good, fmt)
    assert enc.encode2(wild) == want_w and enc.encode2(good) == want_g          # the flag does not stick
    # phantom coefficient through the wide path
    Wp = np.arange(enc.info.block_planes, dtype=np.int32) * 1000003 % 77777
    assert enc.encode2(wild, W=Wp) == oracle.encode(wild, fmt, W=Wp)
    # batch ending on the host: only the refused frame takes the detour
    d = enc.upload(np.stack([good, wild, good]))
    assert enc.encode_batch_to_host(d) == [want_g, want_w, want_g]
    # packets staying in HBM: no host in the loop -> the status says ERANGE
    pk, sizes, status = enc.encode_batch_device(d)
    assert status.cpu().tolist() == [0, -34, 0]
    # the ring, pageable and page-locked
    enc.ring_open(3)
    pin = enc.pinned_frames(1)
    pin[0] = wild
    assert enc.ring_send(good, tag=0) and enc.ring_send(wild, tag=1) and enc.ring_send(pin[0], tag=2, pinned=True)
    assert [enc.ring_receive() for _ in range(3)] == [(0, want_g), (1, want_w), (2, want_w)]
    enc.ring_close()
    enc.free_pinned()
    enc.close()


def test_yuv420_luma_above_depth(oracle):
    """The 4:2:0 front end clips chroma to the depth (output.c:333-393), luma is copied: a wild luma sample
    reaches the T-stage and is coded like the reference would."""
    H, W, depth, fmt = 130, 200, 10, "yuv444p10le"
    enc = _enc(W, H, fmt)
    rng = np.random.default_rng(8)
    y = rng.integers(0, 1024, (H, W)).astype("<u2")
    u = rng.integers(0, 65536, (65, 100)).astype("<u2")       # chroma beyond the depth too: the scaler clips it
    v = rng.integers(0, 1024, (65, 100)).astype("<u2")
    y[17, 23] = 40000
    want = oracle.encode(oracle.sws_420_to_444(y, u, v, depth), fmt)
    assert enc.encode2_420(y, u, v) == want
    enc.ring_open(2)
    assert enc.ring_send_420(y, u, v, tag=4)
    assert enc.ring_receive() == (4, want)
    enc.ring_close()
    enc.close()


@pytest.mark.parametrize("qp", [4, 16])
def test_wide_path_at_qp_above_zero(oracle, qp):
    """global_quality > 0 on frames above their depth: wide T-stage -> PVQ search -> host coder, through encode2, the
    batch entry and send_frame / receive_packet; a well-formed neighbour in the same batch is untouched.  (Parity
    unpinned like all of qp > 0.)"""
    import ctypes as C
    from ffmpeg_ffv2_amd import _lib
    from tests.codec_ctypes import Packet, frame_of, make_ctx
    H, W, fmt, depth = 130, 200, "yuv444p10le", 10
    enc = _enc(W, H, fmt, max_batch=3)
    good = synth.noise(3, 3, H, W, depth)
    wild = want_w = None
    for seed in range(40):                        # noise with a handful of samples above the depth that the reference codes
        cand = synth.noise(100 + seed, 3, H, W, depth).copy()   # without aborting (daala_entropy.c:336) at this qp
        idx = np.random.default_rng(seed).integers(0, cand.size, 12)
        cand.reshape(-1)[idx] = np.random.default_rng(seed + 1).integers(1 << depth, 65536, 12)
        try:
            want_w = oracle.encode(cand, fmt, qp=qp)
            wild = cand
            break
        except RuntimeError:
            continue
    assert wild is not None
    want_g = oracle.encode(good, fmt, qp=qp)
    assert enc.encode2(wild, qp=qp) == want_w
    Wp = (np.arange(enc.info.block_planes, dtype=np.int32) * 7919) % 50
    assert enc.encode2(wild, qp=qp, W=Wp) == oracle.encode(wild, fmt, qp=qp, W=Wp)
    assert enc.encode_batch_to_host(enc.upload(np.stack([good, wild, good])), qp=qp) == [want_g, want_w, want_g]
    enc.close()
    lib = _lib.load()
    ctx = make_ctx(W, H, 70, qp=qp)
    assert lib.ffv2amd_codec_init(C.byref(ctx)) == 0
    for n, fr in enumerate((good, wild)):
        assert lib.ffv2amd_codec_send_frame(C.byref(ctx), C.byref(frame_of(fr, 40 + n)), 0) == 0
    for n, want in enumerate((want_g, want_w)):
        pkt = Packet()
        assert lib.ffv2amd_codec_receive_packet(C.byref(ctx), C.byref(pkt), 1) == 0
        assert pkt.pts == 40 + n and bytes(pkt.data[: pkt.size]) == want
        lib.ffv2amd_packet_unref(C.byref(pkt))
    assert lib.ffv2amd_codec_close(C.byref(ctx)) == 0
