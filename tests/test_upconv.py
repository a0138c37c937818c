"""4:2:0 -> 4:4:4 front end: the reference TOOL's auto-inserted bicubic
scale filter (fftools/ffmpeg_filter.c:63-131, libswscale/utils.c:332-727, swscale.c:96-139,
output.c:333-393).  PARITY UNPINNED: no libswscale binary or vector exists in this environment;
the HIP kernel is checked against oracle/ffv2_swscale_oracle.c, an independent restatement, and
the oracle against properties the reference code implies."""
import numpy as np
import pytest

from ffmpeg_ffv2_amd import frames as synth


def _yuv420(seed, h, w, depth, kind="noise"):
    rng = np.random.default_rng(seed)
    dt = np.uint8 if depth == 8 else np.dtype("<u2")
    ch, cw = (h + 1) // 2, (w + 1) // 2
    if kind == "noise":
        return [rng.integers(0, 1 << depth, s).astype(dt) for s in ((h, w), (ch, cw), (ch, cw))]
    yy, xx = np.mgrid[0:ch, 0:cw]
    ramp = ((3 * xx + 5 * yy + seed) % (1 << depth)).astype(dt)
    return [rng.integers(0, 1 << depth, (h, w)).astype(dt), ramp, ramp[::-1].copy()]


# ---- oracle properties (CPU) ----
def test_filter_rows_are_normalised_and_in_range(oracle):
    for n, one in ((16, 1 << 14), (17, 1 << 12), (240, 1 << 12), (3840, 1 << 14), (8, 1 << 14)):
        f, p = oracle.sws_chroma_filter(n, one)
        taps = f.shape[1]
        assert taps == (4 if n >= 12 else min(5, (n + 1) // 2 - 2))   # utils.c:418-424: min(1 + 4, srcW - 2)
        assert (f.sum(1) == one).all()                       # initFilter normalises every row exactly
        src_n = (n + 1) // 2
        assert (p >= 0).all() and (p + taps <= src_n).all()  # the border fix keeps every tap inside
        assert (np.diff(p) >= 0).all()                       # monotone positions (the scaler's core needs it)
    f, _ = oracle.sws_chroma_filter(3840, 1 << 14)
    # interior: the two phases of a centred 2x bicubic (B = 0, C = 0.6), mirror images of each other
    assert (f[100] == f[101][::-1]).all() or (f[101] == f[102][::-1]).all()


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_luma_identity_and_flat_chroma(oracle, depth):
    y, u, v = _yuv420(3, 37, 50, depth)
    u[:] = 77 % (1 << depth)
    v[:] = (1 << depth) - 1
    out = oracle.sws_420_to_444(y, u, v, depth)
    assert (out[0] == y).all()
    assert (out[1] == u[0, 0]).all() and (out[2] == (1 << depth) - 1).all()      # a flat plane stays flat, no overshoot


def test_oracle_is_separable_and_sample_centred(oracle):
    # chroma constant along x -> the output is too, and equals the 1-D vertical result
    h, w, depth = 32, 24, 8
    y = np.zeros((h, w), np.uint8)
    col = (np.arange(16) * 13 % 256).astype(np.uint8)
    u = np.repeat(col[:, None], 12, 1)
    out = oracle.sws_420_to_444(y, u, u, depth)
    assert (out[1] == out[1][:, :1]).all()
    out_t = oracle.sws_420_to_444(np.zeros((w, h), np.uint8), u.T.copy(), u.T.copy(), depth)
    # same kernel both ways (14- vs 12-bit coefficients differ by rounding only): within 1 LSB
    assert np.abs(out_t[1].T.astype(int) - out[1].astype(int)).max() <= 1


# ---- HIP vs oracle (GPU) ----
CASES = [(8, 240, 320), (10, 128, 192), (12, 130, 200), (8, 65, 129), (10, 37, 51), (8, 16, 16), (12, 1080, 1920)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth,h,w", CASES)
@pytest.mark.parametrize("kind", ["noise", "ramp"])
def test_upconvert_matches_oracle(oracle, depth, h, w, kind):
    from ffmpeg_ffv2_amd import FFV2Encoder
    fmt = {8: "yuv444p", 10: "yuv444p10le", 12: "yuv444p12le"}[depth]
    enc = FFV2Encoder(w, h, fmt, device=0, max_batch=1)
    y, u, v = _yuv420(11, h, w, depth, kind)
    got = enc.upconvert_420(y, u, v)
    want = oracle.sws_420_to_444(y, u, v, depth)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert len(bad) == 0, "first mismatch at (plane, y, x) = %s: %d vs %d" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("depth,h,w", [(8, 240, 320), (10, 270, 480), (12, 135, 240)])
def test_encode_frame_420_equals_convert_then_encode(oracle, depth, h, w):
    """The literal BASELINE pixel formats (yuv420p / 10le / 12le) end to end: 4:2:0 host frame ->
    packet == oracle conversion followed by the oracle encoder (qp 0)."""
    from ffmpeg_ffv2_amd import FFV2Encoder
    fmt = {8: "yuv444p", 10: "yuv444p10le", 12: "yuv444p12le"}[depth]
    enc = FFV2Encoder(w, h, fmt, device=0, max_batch=1)
    for seed in range(2):
        y, u, v = _yuv420(seed, h, w, depth)
        assert enc.encode2_420(y, u, v) == oracle.encode(oracle.sws_420_to_444(y, u, v, depth), fmt)
    enc.close()


@pytest.mark.gpu
def test_420_front_end_needs_a_yuv444_encoder():
    from ffmpeg_ffv2_amd import FFV2Encoder
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = FFV2Encoder(64, 64, "gbrp", device=0)
    with pytest.raises(FFV2Error) as ei:
        enc.encode2_420(np.zeros((64, 64), np.uint8), np.zeros((32, 32), np.uint8), np.zeros((32, 32), np.uint8))
    assert ei.value.code == -22
    enc.close()
