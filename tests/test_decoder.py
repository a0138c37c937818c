"""Decoder side (SURVEY.md 8(f) rank 2 completed): the reference decoder's entropy
layer + dequant_block (ffv2dec.c:76-136 over daala_entropy.c:273-326,413-425) restated in the oracle (C)
and behind the C-ABI (host parse + device scaling + ffv2_inverse.hip), and the FATE-style 4-line
enc/dec report of tools/fate_report.py against the oracle-generated fixtures in tests/golden/fate/.
PARITY UNPINNED: the reference holds no FFV2 decode hash; the decoder's quirks (mag / sqrt(0) at qp 0,
stale pulse slots between bands, the DEBUGGING grid) are reproduced as read from ffv2dec.c."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ffmpeg_ffv2_amd import frames as synth
from tests import packet_parser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = packet_parser.BANDS_START
CASES = [("gray", 1, 64, 64, 8), ("yuv444p", 3, 100, 150, 8), ("yuv444p10le", 3, 130, 200, 10), ("gbrp12le", 3, 70, 129, 12)]


def _expected_coefficients(pkt, nsb, P):
    """dequant_block from the independent Python parse of the entropy layer (tests/packet_parser.py),
    float32 arithmetic spelt out with numpy, the one pulses[] array per block-plane carried across bands."""
    pp = packet_parser.parse_packet(pkt, nsb, P)
    out = np.zeros((nsb * P, 4096), np.int64)
    for bp, (c0, gains, pulses) in enumerate(pp["blocks"]):
        out[bp, 0] = c0
        slot = np.zeros(4097, np.int64)
        for b in range(13):
            lo, ln = 1 + BS[b], BS[b + 1] - BS[b]
            band = np.array(pulses[b], np.int64)
            slot[: band.size] = band
            cnt = int((band * band).sum())
            mag = np.float32(np.float64(np.float32(gains[b])) ** 1.5)
            with np.errstate(divide="ignore", invalid="ignore"):
                mag = np.float32(np.float64(mag) / np.sqrt(np.float64(cnt)))
                v = slot[:ln].astype(np.float32) * mag
            ok = np.isfinite(v) & (v > -2147483904.0) & (v < 2147483648.0)
            iv = np.where(ok, np.trunc(np.where(ok, v, 0)), -2147483648).astype(np.int64)
            n = min(ln, 4096 - lo)
            out[bp, lo: lo + n] = iv[:n]
    return out, pp["qp"]


@pytest.mark.parametrize("fmt,P,H,W,depth", CASES)
@pytest.mark.parametrize("qp", [0, 4, 16, 64])
def test_oracle_dequant_matches_the_python_parse(oracle, fmt, P, H, W, depth, qp):
    fr = synth.noise(11 + qp, P, H, W, depth)
    pkt = oracle.encode(fr, fmt, qp=qp)
    coef, q = oracle.decode_coefficients(pkt, fmt, H, W)
    nsb = ((W + 63) // 64) * ((H + 63) // 64)
    want, q2 = _expected_coefficients(pkt, nsb, P)
    assert q == q2 == qp
    assert np.array_equal(coef.astype(np.int64), want)
    if qp == 0:                                   # ffv2dec.c:134-136: gain / sqrt(0) -> NaN or inf -> 0x80000000
        assert (coef[:, 1:] == -2147483648).all()
    else:
        assert (np.abs(coef[:, 1:]) < (1 << 24)).all()


def test_oracle_decoder_carries_stale_pulses_between_bands(oracle):
    """Band b's unread slots (its loop stops at qp pulses) keep what an earlier band left there and are scaled
    by band b's magnitude (ffv2dec.c:103,118-136).  Visible wherever a band stops early behind a longer one."""
    fr = synth.noise(5, 1, 64, 64, 8)
    pkt = oracle.encode(fr, "gray", qp=4)
    pp = packet_parser.parse_packet(pkt, 1, 1)
    c0, gains, pulses = pp["blocks"][0]
    coef, _ = oracle.decode_coefficients(pkt, "gray", 64, 64)
    reads = [len(p) for p in pulses]
    assert any(reads[b] < BS[b + 1] - BS[b] and max(reads[:b], default=0) > reads[b] for b in range(1, 13)), reads
    want, _ = _expected_coefficients(pkt, 1, 1)
    assert np.array_equal(coef.astype(np.int64), want)
    fresh = want.copy()                            # what a decoder that cleared pulses[] per band would give
    for b in range(13):
        fresh[0, 1 + BS[b] + reads[b]: min(1 + BS[b + 1], 4096)] = 0
    assert not np.array_equal(fresh, want)


def test_oracle_decoder_refuses_damaged_packets(oracle):
    fr = synth.noise(2, 3, 100, 150, 8)
    pkt = bytearray(oracle.encode(fr, "yuv444p", qp=16))
    with pytest.raises(RuntimeError):
        oracle.decode(bytes(pkt), "gray", 100, 150)             # pix_fmt in the header disagrees
    ok = 0
    for cut in (3, 40, len(pkt) // 2):
        try:
            oracle.decode(bytes(pkt[:cut]), "yuv444p", 100, 150)   # truncated: an error or garbage, never a crash
        except RuntimeError:
            ok += 1
    assert ok >= 1


def test_fate_report_c1_fixtures_are_the_oracles(oracle):
    from tools import fate_report
    for name in ("ffv2-C1-qp0", "ffv2-C1-qp16"):
        assert fate_report.run_oracle(name) == open(os.path.join(fate_report.GOLDEN, name)).read(), name


def test_grid_is_the_only_difference(oracle):
    fr = synth.noise(4, 3, 130, 200, 10)
    pkt = oracle.encode(fr, "yuv444p10le", qp=16)
    a, _ = oracle.decode(pkt, "yuv444p10le", 130, 200)
    g, _ = oracle.decode(pkt, "yuv444p10le", 130, 200, grid=True)
    y, x = np.mgrid[0:130, 0:200]
    on = (x % 64 == 0) | (y % 64 == 0)
    assert np.array_equal(a[:, ~on], g[:, ~on])
    assert (g[0][on] == 0).all() and (g[1][on] == 512).all() and (g[2][on] == 512).all()


@pytest.mark.parametrize("fmt,P,H,W,depth", CASES)
@pytest.mark.parametrize("qp", [0, 4, 16, 64])
def test_capi_packet_parse_matches_oracle_without_a_gpu(oracle, fmt, P, H, W, depth, qp):
    """ffv2amd_parse_packet, the host half of ffv2amd_decode_frame (range decoder + dequant_block's symbol order,
    band scales with the host's pow / sqrt): scaled and stored the way the device kernel does it, the coefficients
    equal the oracle decoder's."""
    import ctypes as C
    from ffmpeg_ffv2_amd import _lib
    from tests.oracle_lib import PIX
    lib = _lib.load()
    pkt = None
    for seed in range(21 + qp, 40 + qp):          # a noise frame the reference does not abort on
        try:
            pkt = oracle.encode(synth.noise(seed, P, H, W, depth), fmt, qp=qp)
            break
        except RuntimeError:
            continue
    assert pkt is not None
    nb = ((W + 63) // 64) * ((H + 63) // 64) * P
    pulses = np.zeros((nb, 4096), np.int16)
    mag = np.zeros((nb, 13), np.float32)
    c0 = np.zeros(nb, np.int32)
    pf, q = C.c_int(-1), C.c_int(-1)
    buf = np.frombuffer(pkt, np.uint8)
    assert lib.ffv2amd_parse_packet(buf.ctypes.data_as(C.c_void_p), buf.size, W, H, C.byref(pf), C.byref(q),
                                    pulses.ctypes.data_as(C.c_void_p), mag.ctypes.data_as(C.c_void_p),
                                    c0.ctypes.data_as(C.c_void_p)) == 0
    assert pf.value == PIX[fmt] and q.value == qp
    band = np.searchsorted(np.array(BS[1:13]), np.arange(4095), side="right")        # band of coding index 1 + t
    with np.errstate(invalid="ignore", over="ignore"):
        v = pulses[:, 1:].astype(np.float32) * mag[:, band]
    ok = np.isfinite(v) & (v > -2147483904.0) & (v < 2147483648.0)
    coef = np.where(ok, np.trunc(np.where(ok, v, 0)), -2147483648).astype(np.int64)
    want, _ = oracle.decode_coefficients(pkt, fmt, H, W)
    assert np.array_equal(c0, want[:, 0]) and np.array_equal(coef, want[:, 1:].astype(np.int64))
    # damaged input is refused, not crashed on
    assert lib.ffv2amd_parse_packet(buf.ctypes.data_as(C.c_void_p), 2, W, H, None, None, pulses.ctypes.data_as(C.c_void_p),
                                    mag.ctypes.data_as(C.c_void_p), c0.ctypes.data_as(C.c_void_p)) in (0, -22)


# ---- GPU: the C-ABI's decoder-side check against the oracle's decoder ----
@pytest.mark.gpu
@pytest.mark.parametrize("fmt,P,H,W,depth", CASES + [("yuv444p", 3, 240, 320, 8)])
@pytest.mark.parametrize("qp", [0, 4, 16, 64])
def test_device_decode_matches_oracle(oracle, fmt, P, H, W, depth, qp):
    from ffmpeg_ffv2_amd import FFV2Encoder
    enc = FFV2Encoder(W, H, fmt, device=0, max_batch=1)
    for seed in range(2):
        fr = synth.noise(100 * seed + qp, P, H, W, depth) if (qp or seed) else synth.make("S1", 3, P, H, W, depth)
        pkt = enc.encode2(fr, qp=qp)
        assert pkt == oracle.encode(fr, fmt, qp=qp)
        for grid in (False, True):
            got, q = enc.decode(pkt, grid=grid)
            want, q2 = oracle.decode(pkt, fmt, H, W, grid=grid)
            assert q == q2 == qp
            assert np.array_equal(got, want), (seed, grid)
    enc.close()


@pytest.mark.gpu
def test_device_decode_refuses_foreign_packets(oracle):
    from ffmpeg_ffv2_amd import FFV2Encoder
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = FFV2Encoder(150, 100, "yuv444p", device=0)
    pkt = oracle.encode(synth.noise(1, 1, 100, 150, 8), "gray", qp=0)
    with pytest.raises(FFV2Error) as ei:
        enc.decode(pkt)
    assert ei.value.code == -22
    with pytest.raises(FFV2Error):
        enc.decode(b"\x00\x01")
    enc.close()


@pytest.mark.gpu
def test_fate_report_gpu_equals_fixtures():
    """tools/fate_report.py through the C-ABI (encode2 + decoder-side check) reproduces the oracle-generated
    4-line reports of every case, C3 (3840x2160 10-bit) at qp 16 included: FATE's own pass criterion, a diff."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fate_report.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    from tools import fate_report
    for name in fate_report.case_names():
        assert open(os.path.join(fate_report.GOLDEN, name)).read() in r.stdout, name
