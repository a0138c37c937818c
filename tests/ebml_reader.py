"""Independent EBML / Matroska reader for the tests (RFC 8794 element ids and sizes, the
Matroska element table for the handful of ids the muxer may emit).  Not the writer's code path."""
import struct

MASTERS = {0x1A45DFA3: "EBML", 0x18538067: "Segment", 0x1549A966: "Info", 0x1654AE6B: "Tracks", 0xAE: "TrackEntry",
           0xE0: "Video", 0x1F43B675: "Cluster", 0x114D9B74: "SeekHead", 0x4DBB: "Seek"}
UINTS = {0x4286: "EBMLVersion", 0x42F7: "EBMLReadVersion", 0x42F2: "EBMLMaxIDLength", 0x42F3: "EBMLMaxSizeLength",
         0x4287: "DocTypeVersion", 0x4285: "DocTypeReadVersion", 0x2AD7B1: "TimecodeScale", 0xD7: "TrackNumber",
         0x73C5: "TrackUID", 0x9C: "FlagLacing", 0x83: "TrackType", 0x23E383: "DefaultDuration", 0xB0: "PixelWidth",
         0xBA: "PixelHeight", 0x54B2: "DisplayUnit", 0xE7: "Timecode", 0x53AC: "SeekPosition"}
STRINGS = {0x4282: "DocType", 0x4D80: "MuxingApp", 0x5741: "WritingApp", 0x22B59C: "Language", 0x86: "CodecID"}
FLOATS = {0x4489: "Duration"}
BINARY = {0xA3: "SimpleBlock", 0x63A2: "CodecPrivate", 0xEC: "Void", 0x53AB: "SeekID"}


def read_id(b, p):
    first = b[p]
    n = 1
    while n <= 4 and not first & (0x80 >> (n - 1)):
        n += 1
    if n > 4:
        raise ValueError("bad element id at %d" % p)
    return int.from_bytes(b[p:p + n], "big"), p + n


def read_size(b, p):
    first = b[p]
    n = 1
    while n <= 8 and not first & (0x80 >> (n - 1)):
        n += 1
    if n > 8:
        raise ValueError("bad element size at %d" % p)
    v = int.from_bytes(b[p:p + n], "big") & ((1 << (7 * n)) - 1)
    if v == (1 << (7 * n)) - 1:
        raise ValueError("unknown-size element at %d" % p)
    return v, p + n


def parse(b, p=0, end=None):
    """-> list of (name, value) ; masters carry a nested list; unknown ids raise."""
    end = len(b) if end is None else end
    out = []
    while p < end:
        eid, p = read_id(b, p)
        size, p = read_size(b, p)
        if p + size > end:
            raise ValueError("element 0x%X overruns its parent" % eid)
        body = b[p:p + size]
        if eid in MASTERS:
            out.append((MASTERS[eid], parse(b, p, p + size)))
        elif eid in UINTS:
            out.append((UINTS[eid], int.from_bytes(body, "big")))
        elif eid in STRINGS:
            out.append((STRINGS[eid], body.decode("ascii")))
        elif eid in FLOATS:
            out.append((FLOATS[eid], struct.unpack(">d" if size == 8 else ">f", body)[0]))
        elif eid in BINARY:
            out.append((BINARY[eid], bytes(body)))
        else:
            raise ValueError("unexpected element id 0x%X" % eid)
        p += size
    return out


def child(tree, name):
    hits = [v for n, v in tree if n == name]
    assert len(hits) == 1, (name, len(hits))
    return hits[0]


def blocks(segment):
    """-> [(absolute timestamp in TimecodeScale units, track, flags, payload)] in file order"""
    out = []
    for n, cl in segment:
        if n != "Cluster":
            continue
        assert cl[0][0] == "Timecode"            # the cluster timestamp comes first
        base = cl[0][1]
        for m, v in cl[1:]:
            assert m == "SimpleBlock"
            track, q = read_size(v, 0)
            rel = struct.unpack(">h", v[q:q + 2])[0]
            out.append((base + rel, track, v[q + 2], v[q + 3:]))
    return out
