"""Matroska wire step: the files are read back with an independent EBML
reader.  PARITY UNPINNED against libavformat's bytes (no FFmpeg binary here); what is pinned is
what the reference's demuxer keys on: DocType, the "V_FFV2" codec id of libavformat/matroska.c:83,
one video track without CodecPrivate, SimpleBlocks without the keyframe bit, payload bytes intact."""
import os

import numpy as np
import pytest

from tests import ebml_reader as E

from ffmpeg_ffv2_amd import _lib
from ffmpeg_ffv2_amd.mkv import MkvWriter


def _lib_or_skip():
    try:
        return _lib.load()
    except Exception as exc:        # the library is built by __graft_entry__.build()
        pytest.skip("libffv2amd.so not built: %s" % exc)


def _roundtrip(tmp_path, packets, fps=(25, 1), w=320, h=240, pts=None):
    _lib_or_skip()
    path = tmp_path / "t.mkv"
    with MkvWriter(path, w, h, fps) as m:
        for i, p in enumerate(packets):
            m.write(p, None if pts is None else pts[i])
    data = path.read_bytes()
    tree = E.parse(data)
    return data, tree


def test_header_tracks_and_payloads(tmp_path):
    rng = np.random.default_rng(5)
    packets = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in (1, 130, 17000, 185019, 4)]
    data, tree = _roundtrip(tmp_path, packets)
    assert [n for n, _ in tree] == ["EBML", "Segment"]
    ebml = dict(tree[0][1])
    assert ebml == {"EBMLVersion": 1, "EBMLReadVersion": 1, "EBMLMaxIDLength": 4, "EBMLMaxSizeLength": 8,
                    "DocType": "matroska", "DocTypeVersion": 4, "DocTypeReadVersion": 2}
    seg = tree[1][1]
    # SeekHead + Void fill the 229 bytes the stock muxer reserves (matroskaenc.c:170,439,509-513), Info follows
    assert [n for n, _ in seg][:4] == ["SeekHead", "Void", "Info", "Tracks"]
    void = E.child(seg, "Void")
    assert set(void) <= {0}
    # the SeekHead's positions (relative to the segment's first data byte) land on the elements they name
    seg_id0 = data.index(bytes.fromhex("18538067"))
    _, seg_data = E.read_size(data, seg_id0 + 4)
    seeks = [dict(s) for n, s in E.child(seg, "SeekHead") if n == "Seek"]
    assert [s["SeekID"].hex() for s in seeks] == ["1549a966", "1654ae6b"]
    for s in seeks:
        at = seg_data + s["SeekPosition"]
        assert data[at:at + 4] == s["SeekID"]
    assert seeks[0]["SeekPosition"] == 10 * 21 + 19
    info = dict(E.child(seg, "Info"))
    assert info["TimecodeScale"] == 1000000
    assert info["Duration"] == 200.0                       # 5 frames at 25 fps, in ms
    entry = dict(E.child(E.child(seg, "Tracks"), "TrackEntry"))
    assert entry["CodecID"] == "V_FFV2" and entry["TrackType"] == 1 and entry["TrackNumber"] == 1
    assert entry["FlagLacing"] == 0 and entry["Language"] == "und" and entry["DefaultDuration"] == 40000000
    assert "CodecPrivate" not in entry
    assert dict(entry["Video"]) == {"PixelWidth": 320, "PixelHeight": 240, "DisplayUnit": 4}
    got = E.blocks(seg)
    assert [b[3] for b in got] == packets
    assert [b[0] for b in got] == [0, 40, 80, 120, 160]
    assert all(b[1] == 1 and b[2] == 0 for b in got)       # track 1, no keyframe / lacing flags
    # the Segment's size field covers the file to its end
    seg_id = data.index(bytes.fromhex("18538067"))
    size, p = E.read_size(data, seg_id + 4)
    assert p + size == len(data)


def test_cluster_policy_and_ntsc_timestamps(tmp_path):
    packets = [bytes([i & 255]) * 700000 for i in range(20)]            # 14 MB: the 5 MiB cluster limit applies
    _, tree = _roundtrip(tmp_path, packets, fps=(30000, 1001))
    seg = tree[1][1]
    clusters = [v for n, v in seg if n == "Cluster"]
    assert len(clusters) >= 3
    for cl in clusters:
        assert sum(len(v) for n, v in cl if n == "SimpleBlock") <= 5 * 1024 * 1024 + 700004
    got = E.blocks(seg)
    assert [b[3] for b in got] == packets
    assert [b[0] for b in got] == [(2 * i * 1001 * 1000 + 30000) // 60000 for i in range(20)]


def test_sparse_timestamps_open_new_clusters(tmp_path):
    pts = [0, 1, 200, 201, 2000]                                          # 25 fps: 0, 40 ms, 8 s, 8.04 s, 80 s
    _, tree = _roundtrip(tmp_path, [b"a", b"bb", b"ccc", b"dddd", b"eeeee"], pts=pts)
    seg = tree[1][1]
    assert [v[0][1] for n, v in seg if n == "Cluster"] == [0, 8000, 80000]
    assert [b[0] for b in E.blocks(seg)] == [0, 40, 8000, 8040, 80000]
    assert dict(E.child(seg, "Info"))["Duration"] == 80040.0


def test_argument_errors(tmp_path):
    lib = _lib_or_skip()
    import ctypes as C
    h = C.c_void_p()
    lib.ffv2amd_mkv_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    assert lib.ffv2amd_mkv_open(C.byref(h), str(tmp_path / "x.mkv").encode(), 0, 240, 25, 1) == -22
    assert lib.ffv2amd_mkv_open(C.byref(h), str(tmp_path / "no" / "dir.mkv").encode(), 320, 240, 25, 1) == -5
    with MkvWriter(tmp_path / "y.mkv", 64, 64) as m:
        m.write(b"x", 3)
        with pytest.raises(_lib.FFV2Error):
            m.write(b"y", 2)                                              # pts must not decrease
    assert os.path.getsize(tmp_path / "y.mkv") > 0


@pytest.mark.gpu
def test_encoded_packets_survive_the_container(tmp_path):
    from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
    enc = FFV2Encoder(320, 240, "yuv444p10le")
    pk = [enc.encode2(synth.make("S1" if n % 2 == 0 else "S2", n, 3, 240, 320, 10)) for n in range(4)]
    enc.close()
    _, tree = _roundtrip(tmp_path, pk)
    assert [b[3] for b in E.blocks(tree[1][1])] == [bytes(p) for p in pk]


@pytest.mark.gpu
def test_cli_writes_the_same_packets_into_matroska(tmp_path):
    import subprocess
    from ffmpeg_ffv2_amd import frames as synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", root, "examples/ffv2enc_cli"], check=True)
    src = tmp_path / "in.yuv"
    src.write_bytes(b"".join(synth.make("S2", n, 3, 240, 320, 8).tobytes() for n in range(3)))
    cli = os.path.join(root, "examples", "ffv2enc_cli")
    for dst in ("out.ffv2", "out.mkv"):
        subprocess.run([cli, "320", "240", "yuv444p", str(src), str(tmp_path / dst)], check=True, capture_output=True)
    tree = E.parse((tmp_path / "out.mkv").read_bytes())
    got = E.blocks(tree[1][1])
    assert len(got) == 3 and b"".join(b[3] for b in got) == (tmp_path / "out.ffv2").read_bytes()
