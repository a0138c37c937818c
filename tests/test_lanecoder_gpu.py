"""GPU tests of the qp > 0 coder with many frames in flight (ffv2_lanecoder.hip, SURVEY.md 8(f) rank 1,
8/A14): the range coder's serial chain runs one frame per lane, CDF rows / raw bits / carries are
data-parallel.  Packets are held to the host coder's (ffv2amd_encode_batch_to_host) and the CPU
oracle's.  Parity unpinned for qp > 0: the oracle restates the reference's PVQ asm, which cannot be
assembled here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ffmpeg_ffv2_amd import frames as synth  # noqa: E402


def _enc(w, h, fmt, max_batch):
    from ffmpeg_ffv2_amd import FFV2Encoder, build
    build.build()
    return FFV2Encoder(w, h, fmt, device=0, max_batch=max_batch)


@pytest.mark.parametrize("fmt,P,H,W,depth", [("gray", 1, 64, 64, 8), ("yuv444p", 3, 100, 150, 8),
                                             ("yuv444p10le", 3, 130, 200, 10)])
@pytest.mark.parametrize("qp", [2, 4, 16, 33, 64])
def test_lanecoder_matches_host_coder_and_oracle(oracle, fmt, P, H, W, depth, qp):
    enc = _enc(W, H, fmt, 3)
    n = 5                                            # two T-stage batches, one partly filled group of lanes
    frames = np.stack([synth.noise(7 * qp + i, P, H, W, depth) if i % 2 == 0 else synth.make("S1", i, P, H, W, depth)
                       for i in range(n)])
    dev = enc.upload(frames)
    enc.lanecoder_open(n)
    pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
    for i in range(n):
        try:
            want = oracle.encode(frames[i], fmt, qp=qp)
        except Exception:
            want = None                              # the reference would av_assert0 on this frame
        if want is None:
            assert status[i] == -1, (i, status[i])
        else:
            assert status[i] == 0, (i, status[i])
            assert pk[i, : sizes[i]].tobytes() == want, i
    enc.lanecoder_close()
    enc.close()


def test_lanecoder_more_frames_than_lanes(oracle):
    """70 small frames: two groups of lanes (64 + 6), frames of different lengths side by side."""
    W, H, fmt, P, depth, qp = 64, 128, "gray", 1, 8, 8
    enc = _enc(W, H, fmt, 16)
    n = 70
    frames = np.stack([synth.noise(1000 + i, P, H, W, depth) if i % 3 else synth.make("S2", i, P, H, W, depth)
                       for i in range(n)])
    # frames with less to code: a quiet lower half
    frames[5, :, 64:, :] = 120 + frames[5, :, 64:, :] % 16
    frames[66, :, 32:, :] = 7 + frames[66, :, 32:, :] % 3
    dev = enc.upload(frames)
    enc.lanecoder_open(n)
    pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
    lengths = set()
    for i in range(n):
        try:
            want = oracle.encode(frames[i], fmt, qp=qp)
        except Exception:
            want = None
        if want is None:
            assert status[i] == -1, i
        else:
            assert status[i] == 0 and pk[i, : sizes[i]].tobytes() == want, i
            lengths.add(len(want))
    assert len(lengths) > 3
    good = [i for i in range(16) if status[i] == 0]
    if len(good) == 16:
        assert enc.encode_batch_to_host(dev[:16], qp=qp) == [pk[i, : sizes[i]].tobytes() for i in range(16)]
    # a second call reuses the scratch (raw-bit tail and flags are cleared per call)
    first = [pk[i, : sizes[i]].tobytes() for i in range(3)]
    pk2, sizes2, status2 = enc.lanecoder_encode(dev[:3], qp, as_arrays=True)
    assert [pk2[i, : sizes2[i]].tobytes() for i in range(3)] == first and list(status2) == list(status[:3])
    enc.close()


def test_lanecoder_abort_and_arguments(oracle):
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = _enc(64, 64, "gray", 2)
    flat = np.full((2, 1, 64, 64), 200, np.uint8)
    flat[0, 0, 10, 10] = 0                           # pulses concentrate: daala_entropy.c:336
    flat[1] = synth.noise(3, 1, 64, 64, 8)
    dev = enc.upload(flat)
    with pytest.raises(FFV2Error):                   # not opened
        enc.lanecoder_encode(dev, 16)
    enc.lanecoder_open(2)
    pk, sizes, status = enc.lanecoder_encode(dev, 16, as_arrays=True)
    assert status[0] == -1 and status[1] == 0
    assert pk[1, : sizes[1]].tobytes() == oracle.encode(flat[1], "gray", qp=16)
    # qp 1 always aborts in the reference (ft = 1, daala_entropy.c:342)
    pk, sizes, status = enc.lanecoder_encode(dev, 1, as_arrays=True)
    assert list(status) == [-1, -1]
    with pytest.raises(FFV2Error):
        enc.lanecoder_encode(dev, 65)
    enc.close()


def test_lanecoder_random_geometries_against_host_coder():
    """Random sizes / formats / qp / content mixes: every frame's packet (or abort) must equal the
    host coder's, frame by frame."""
    import random
    from ffmpeg_ffv2_amd._lib import FFV2Error
    rnd = random.Random(20261004)
    fmts = [("gray", 1, 8), ("yuv444p", 3, 8), ("yuv444p10le", 3, 10), ("gbrp12le", 3, 12), ("yuv444p12le", 3, 12)]
    checked = aborted = 0
    for it in range(24):
        fmt, P, depth = rnd.choice(fmts)
        W, H = rnd.randint(1, 260), rnd.randint(1, 200)
        qp = rnd.choice([2, 3, 5, 8, 16, 17, 32, 48, 64])
        n = rnd.randint(1, 7)
        enc = _enc(W, H, fmt, rnd.randint(1, 3))
        kinds = [rnd.choice(["S2", "S2", "S1", "flatnoise"]) for _ in range(n)]
        frames = []
        for i, k in enumerate(kinds):
            if k == "flatnoise":
                f = synth.noise(it * 10 + i, P, H, W, depth)
                f = (f % 5 + (1 << (depth - 1))).astype(f.dtype)
            else:
                f = synth.make(k, it * 10 + i, P, H, W, depth)
            frames.append(f)
        frames = np.stack(frames)
        dev = enc.upload(frames)
        enc.lanecoder_open(n)
        pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
        for i in range(n):
            try:
                want = enc.encode_batch_to_host(dev[i:i + 1], qp=qp)[0]
            except FFV2Error as e:
                assert status[i] == e.code, (it, i, status[i], e.code)
                aborted += 1
                continue
            assert status[i] == 0 and pk[i, : sizes[i]].tobytes() == want, (it, i, fmt, W, H, qp, kinds[i])
            checked += 1
        enc.close()
    assert checked > 40 and aborted > 0


def test_lanecoder_two_calls_in_flight(oracle):
    """submit(n+1) before finish(n): the front of call n+1 (own buffers) runs beside the chain of call n;
    a third submit is refused; packets come back per call, identical to the synchronous path."""
    W, H, fmt, P, depth, qp = 200, 130, "yuv444p", 3, 8, 16
    enc = _enc(W, H, fmt, 2)
    calls = [np.stack([synth.noise(100 * c + i, P, H, W, depth) for i in range(5 - c)]) for c in range(3)]
    dev = [enc.upload(c) for c in calls]
    enc.lanecoder_open(5)
    want = []
    for c in range(3):
        pk, sizes, status = enc.lanecoder_encode(dev[c], qp, as_arrays=True)
        assert not status.any()
        want.append([pk[i, : sizes[i]].tobytes() for i in range(len(calls[c]))])
    assert want[0][0] == oracle.encode(calls[0][0], fmt, qp=qp)
    assert enc.lanecoder_submit(dev[0], qp) and enc.lanecoder_submit(dev[1], qp)
    assert not enc.lanecoder_submit(dev[2], qp)         # two in flight already: EAGAIN
    got = []
    pk, sizes, status = enc.lanecoder_finish()
    got.append([pk[i, : sizes[i]].tobytes() for i in range(len(calls[0]))])
    assert enc.lanecoder_submit(dev[2], qp)
    for c in (1, 2):
        pk, sizes, status = enc.lanecoder_finish()
        assert not status.any()
        got.append([pk[i, : sizes[i]].tobytes() for i in range(len(calls[c]))])
    assert got == want
    # three calls in flight: no finish is due between the submits
    enc.lanecoder_open(5, calls_in_flight=3)
    assert all(enc.lanecoder_submit(dev[c], qp) for c in range(3)) and not enc.lanecoder_submit(dev[0], qp)
    for c in range(3):
        pk, sizes, status = enc.lanecoder_finish()
        assert not status.any() and [pk[i, : sizes[i]].tobytes() for i in range(len(calls[c]))] == want[c], c
    # the same packets as they lie on the device, in one copy
    assert enc.lanecoder_submit(dev[0], qp)
    buf, offs, sizes, status = enc.lanecoder_finish_packed()
    assert not status.any() and all(int(o) % 16 == 0 for o in offs)
    assert [buf[int(o): int(o) + int(n)].tobytes() for o, n in zip(offs, sizes)] == want[0]
    enc.close()


@pytest.mark.parametrize("calls,backs", [(2, 2), (3, 2), (4, 4), (4, 1)])
def test_lanecoder_range_chains_side_by_side(oracle, calls, backs):
    """ffv2amd_lanecoder_open_ex: `backs` range chains at a time (call n on back n % backs, each with its own records,
    code words and lane state).  Calls of different content and length, every set and back reused several times, small
    windows so that every call's back is many cdf / chain launches: packets equal the synchronous path's; the
    memory estimate grows by a back's share; arguments out of range are refused."""
    from ffmpeg_ffv2_amd import _lib
    W, H, fmt, P, depth, qp = 200, 130, "yuv444p10le", 3, 10, 16
    enc = _enc(W, H, fmt, 2)
    ncall = 9
    batches = [np.stack([synth.noise(31 * c + i, P, H, W, depth) if (c + i) % 3 else synth.make("S1", c + i, P, H, W, depth)
                         for i in range(1 + (c * 5) % 7)]) for c in range(ncall)]
    dev = [enc.upload(b) for b in batches]
    enc.lanecoder_open(7)
    want = []
    for c in range(ncall):
        pk, sizes, status = enc.lanecoder_encode(dev[c], qp, as_arrays=True)
        assert not status.any()
        want.append([pk[i, : sizes[i]].tobytes() for i in range(len(batches[c]))])
    assert want[3][0] == oracle.encode(batches[3][0], fmt, qp=qp)
    one = enc.lanecoder_bytes_per_frame(0, calls, 1)
    assert enc.lanecoder_bytes_per_frame(0, calls, backs) > one or backs == 1
    lib = _lib.load()
    lib.ffv2amd_debug_lanecoder_window(4096)
    try:
        enc.lanecoder_open(7, calls_in_flight=calls, backs=backs)
    finally:
        lib.ffv2amd_debug_lanecoder_window(0)
    sub = fin = 0
    while fin < ncall:
        while sub < ncall and sub - fin < calls:
            assert enc.lanecoder_submit(dev[sub], qp)
            sub += 1
        if sub < ncall:
            assert not enc.lanecoder_submit(dev[sub], qp)   # every set is in flight
        pk, sizes, status = enc.lanecoder_finish()
        assert not status.any()
        assert [pk[i, : sizes[i]].tobytes() for i in range(len(batches[fin]))] == want[fin], fin
        fin += 1
    for bad in [(2, 3), (5, 1), (1, 1), (3, -1)]:
        with pytest.raises(_lib.FFV2Error):
            enc.lanecoder_open(7, calls_in_flight=bad[0], backs=bad[1])
    enc.close()


def test_lanecoder_small_packet_cap_reports_nospace(oracle):
    """A tight packet_cap saves HBM; a frame that does not fit is refused, the others are unaffected."""
    W, H, fmt, P, depth, qp = 128, 128, "gray", 1, 8, 16
    enc = _enc(W, H, fmt, 2)
    quiet = np.full((P, H, W), 128, np.uint8)
    quiet[:, :64, :64] = synth.noise(2, P, 64, 64, depth)          # one busy block, three flat ones
    frames = np.stack([synth.noise(1, P, H, W, depth), quiet])
    want = [oracle.encode(f, fmt, qp=qp) for f in frames]
    assert len(want[0]) > len(want[1]) + 32                # 670 and 623 bytes
    cap = len(want[1]) + 20                                # rounded up to 16 inside: still short of frame 0
    dev = enc.upload(frames)
    enc.lanecoder_open(2, packet_cap=cap)
    assert enc.lanecoder_bytes_per_frame(cap) < enc.lanecoder_bytes_per_frame()
    pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
    assert status[0] == -28 and status[1] == 0
    assert pk[1, : sizes[1]].tobytes() == want[1]
    enc.lanecoder_open(2)                              # reopen with the default bound
    pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
    assert [pk[i, : sizes[i]].tobytes() for i in range(2)] == want
    enc.close()


def test_lanecoder_phantom_coefficient_w(oracle):
    """the phantom 2049th coefficient of band 12 (W, per block-plane) takes part in the gain
    and the PVQ search; frames of a call beyond the first T-stage batch must pick up their own W."""
    import torch
    W, H, fmt, P, depth, qp, n = 130, 70, "yuv444p", 3, 8, 16, 5
    enc = _enc(W, H, fmt, 2)
    frames = np.stack([synth.noise(50 + i, P, H, W, depth) for i in range(n)])
    Wv = np.random.default_rng(9).integers(-3000, 3000, (n, enc.info.block_planes)).astype(np.int32)
    enc.lanecoder_open(n)
    pk, sizes, status = enc.lanecoder_encode(enc.upload(frames), qp, d_W=torch.from_numpy(Wv).cuda(), as_arrays=True)
    for i in range(n):
        assert status[i] == 0 and pk[i, : sizes[i]].tobytes() == oracle.encode(frames[i], fmt, qp=qp, W=Wv[i]), i
    enc.close()


def test_lanecoder_frame_with_out_of_depth_samples_fails_alone(oracle):
    """A frame the T-stage refuses (sample above the declared depth -> ERANGE, include/ffv2_amd.h) comes back
    with that status; its neighbours in the same group of lanes are coded as usual."""
    W, H, fmt, P, depth, qp = 96, 80, "yuv444p10le", 3, 10, 16
    enc = _enc(W, H, fmt, 2)
    frames = np.stack([synth.noise(70 + i, P, H, W, depth) for i in range(4)])
    frames[2, 1, 40, 50] = 1500                      # 11-bit value in a 10-bit format
    enc.lanecoder_open(4)
    pk, sizes, status = enc.lanecoder_encode(enc.upload(frames), qp, as_arrays=True)
    assert list(status) == [0, 0, -34, 0]
    for i in (0, 1, 3):
        assert pk[i, : sizes[i]].tobytes() == oracle.encode(frames[i], fmt, qp=qp), i
    enc.close()


def test_lanecoder_open_beyond_hbm_is_refused_cleanly(oracle):
    """More frames in flight than HBM holds: FFV2AMD_ERR_NOMEM (-12), nothing leaks into later calls."""
    from ffmpeg_ffv2_amd._lib import FFV2Error
    enc = _enc(1920, 1080, "yuv444p", 1)
    with pytest.raises(FFV2Error) as ei:
        enc.lanecoder_open(100000)                       # 8.4 TB
    assert ei.value.code == -12
    frame = synth.noise(5, 3, 1080, 1920, 8)[None]
    assert enc.encode_batch_to_host(enc.upload(frame), qp=0) == [oracle.encode(frame[0], "yuv444p")]
    enc.lanecoder_open(1)
    pk, sizes, status = enc.lanecoder_encode(enc.upload(frame), 16, as_arrays=True)
    assert status[0] == 0 and pk[0, : sizes[0]].tobytes() == oracle.encode(frame[0], "yuv444p", qp=16)
    enc.close()


@pytest.mark.parametrize("window", [16, 48, 1008, 5000, 0])
@pytest.mark.parametrize("fmt,P,H,W,depth,qp", [("yuv444p", 3, 100, 150, 8, 16), ("gray", 1, 130, 70, 8, 5), ("yuv444p10le", 3, 64, 200, 10, 40)])
def test_lanecoder_windows_of_the_coding_order(oracle, window, fmt, P, H, W, depth, qp):
    """Round 3: cdf and chain alternate over windows of the coding order (records exist for two windows only).
    Whatever the window -- 16 symbols cut every chunk of a CDF row, 48 cut it at changing places, 0 is the
    default of 2^18 -- the packets are the host coder's and the oracle's; frames of uneven length (a quiet
    half, a structured frame that aborts) share the group of lanes."""
    from ffmpeg_ffv2_amd import _lib
    lib = _lib.load()
    lib.ffv2amd_debug_lanecoder_window(window)
    try:
        enc = _enc(W, H, fmt, 2)
        n = 5
        frames = np.stack([synth.noise(31 * qp + i, P, H, W, depth) for i in range(n)])
        frames[1, :, H // 2:, :] = (1 << (depth - 1)) + frames[1, :, H // 2:, :] % 3          # much less to code
        frames[3] = synth.make("S1", 3, P, H, W, depth)                                          # usually aborts
        dev = enc.upload(frames)
        enc.lanecoder_open(n)
        for rep in range(2):                                  # the second call finds the first one's state in the scratch
            pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
            for i in range(n):
                try:
                    want = oracle.encode(frames[i], fmt, qp=qp)
                except Exception:
                    want = None
                if want is None:
                    assert status[i] == -1, (rep, i, status[i])
                else:
                    assert status[i] == 0 and pk[i, : sizes[i]].tobytes() == want, (rep, i, window)
        # two calls in flight: the second call's windows wait for the first call's back
        assert enc.lanecoder_submit(dev[:3], qp) and enc.lanecoder_submit(dev[2:], qp)
        pk_a, sz_a, _ = enc.lanecoder_finish()
        first = pk_a[2, : sz_a[2]].tobytes()               # the page-locked array is reused by the next finish of this shape
        pk_b, sz_b, _ = enc.lanecoder_finish()
        assert first == pk_b[0, : sz_b[0]].tobytes() == oracle.encode(frames[2], fmt, qp=qp)
        enc.close()
    finally:
        lib.ffv2amd_debug_lanecoder_window(0)


@pytest.mark.parametrize("fmt,P,H,W,depth,qp", [("yuv444p", 3, 100, 150, 8, 16), ("gray", 1, 130, 70, 8, 2),
                                                ("yuv444p10le", 3, 64, 200, 10, 64), ("yuv444p12le", 3, 70, 70, 12, 7)])
def test_lanecoder_counts_from_the_search_equal_the_count_pass(oracle, monkeypatch, fmt, P, H, W, depth, qp):
    """Round 3: the PVQ search notes what the coder will read of each band (symbols until the magnitudes reach qp,
    sign bits, a pulse as large as the alphabet -> abort) instead of a kernel walking the pulses again
    (FFV2AMD_LC_COUNT_KERNEL=1 brings that pass back).  Same packets, same aborts, both equal to the oracle."""
    enc = _enc(W, H, fmt, 2)
    n = 6
    frames = np.stack([synth.noise(17 * qp + i, P, H, W, depth) for i in range(n)])
    frames[1, :, H // 2:, :] = (1 << (depth - 1)) + frames[1, :, H // 2:, :] % 3          # bands without a pulse
    frames[2] = synth.make("S1", 2, P, H, W, depth)                                          # usually aborts
    frames[4] = (1 << (depth - 1))                                                           # flat: nothing but zeros
    frames[4, :, 3, 5] += 9
    dev = enc.upload(frames)
    enc.lanecoder_open(n)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FFV2AMD_LC_COUNT_KERNEL", mode)
        pk, sizes, status = enc.lanecoder_encode(dev, qp, as_arrays=True)
        got[mode] = [(int(status[i]), pk[i, : sizes[i]].tobytes() if status[i] == 0 else None) for i in range(n)]
    assert got["0"] == got["1"]
    for i in range(n):
        try:
            want = oracle.encode(frames[i], fmt, qp=qp)
        except Exception:
            want = None
        assert got["0"][i] == ((0, want) if want is not None else (-1, None)), i
    enc.close()
