"""CPU-only checks of the C-ABI library: it loads, exports every symbol the
headers declare, and its host-side arithmetic (no GPU work) agrees with the
oracle.  No compute entry point is called here."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ffmpeg_ffv2_amd import _lib, build
    build.build()
    return _lib.load()


def test_exports_cover_headers(lib):
    declared = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        declared |= set(re.findall(r"\b(ffv2amd_\w+)\s*\(", txt))
    assert len(declared) >= 14
    for name in sorted(declared):
        assert hasattr(lib, name), "include/*.h declares %s but libffv2amd.so lacks it" % name


def test_version(lib):
    assert b"gfx950" in lib.ffv2amd_version()


def test_coded_gain_matches_oracle(lib, oracle):
    rng = np.random.default_rng(7)
    es = np.concatenate([np.arange(0, 4000), rng.integers(0, 1 << 20, 4000),
                         rng.integers(0, 1 << 42, 4000), [2049 * (1 << 40), (64 * 2032) ** 2]])
    for e in es:
        assert lib.ffv2amd_coded_gain(int(e)) == oracle.coded_gain(int(e))
    assert lib.ffv2amd_coded_gain((64 * 2032) ** 2) == 2566        # SURVEY.md KAT: band-0 gain of a white block


@pytest.mark.parametrize("fmt,shape", [("gray", (1, 64, 64)), ("gray", (1, 100, 150)), ("yuv444p", (3, 240, 320)),
                                       ("yuv444p10le", (3, 128, 192)), ("gbrp12le", (3, 200, 700)),
                                       ("yuv444p", (3, 1080, 1920))])
def test_range_prefix_matches_oracle_packet(lib, oracle, fmt, shape):
    """The range-coded head of a qp=0 packet is data independent: compare the
    library's prefix with the head of an oracle packet of a flat frame."""
    from tests.oracle_lib import PIX
    P, H, W = shape
    nsb = ((W + 63) // 64) * ((H + 63) // 64)
    buf = (C.c_uint8 * 65536)()
    slack = C.c_int(-1)
    n = lib.ffv2amd_range_prefix(PIX[fmt], nsb, buf, 65536, C.byref(slack))
    assert n >= 1 and 0 <= slack.value <= 7
    pre = bytes(buf[:n])
    dt = np.uint8 if not fmt.endswith("le") else np.uint16
    mid = 128 if dt == np.uint8 else (1 << (int(re.search(r"(\d+)le", fmt).group(1)) - 1))
    pk = oracle.encode(np.full(shape, mid, dt), fmt)
    assert pk[: n - 1] == pre[: n - 1]
    mask = 0xFF & ~((1 << slack.value) - 1)
    assert pk[n - 1] & mask == pre[n - 1]
    assert pre[n - 1] & ~mask == 0


def test_range_prefix_rejects_bad_args(lib):
    assert lib.ffv2amd_range_prefix(196, 1, None, 0, None) == -22
    assert lib.ffv2amd_range_prefix(5, 0, None, 0, None) == -22


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    r = lib.ffv2amd_encoder_create(C.byref(h), 320, 240, 5, 0, 1)
    assert r == -5 and not h.value          # FFV2AMD_ERR_DEVICE, never a CPU fallback


def test_create_rejects_unsupported_pix_fmt(lib):
    h = C.c_void_p()
    assert lib.ffv2amd_encoder_create(C.byref(h), 320, 240, 0, 0, 1) == -22    # yuv420p: utils.c:814-822
    assert lib.ffv2amd_encoder_create(C.byref(h), 0, 240, 5, 0, 1) == -22
