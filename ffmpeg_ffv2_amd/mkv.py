"""Matroska "V_FFV2" writer (include/ffv2_amd_mkv.h) - the wire step after encode2()."""
import ctypes as C

from . import _lib


class MkvWriter:
    """with MkvWriter(path, width, height, fps=(25, 1)) as m: m.write(packet_bytes, pts)"""

    def __init__(self, path, width, height, fps=(25, 1)):
        self._lib = _lib.load()
        self._lib.ffv2amd_mkv_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        self._lib.ffv2amd_mkv_write_packet.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64]
        self._lib.ffv2amd_mkv_close.argtypes = [C.c_void_p]
        self._h = C.c_void_p()
        _lib.check(self._lib.ffv2amd_mkv_open(C.byref(self._h), str(path).encode(), width, height, fps[0], fps[1]),
                   "ffv2amd_mkv_open")
        self._n = 0

    def write(self, packet, pts=None):
        pts = self._n if pts is None else pts
        packet = bytes(packet)
        _lib.check(self._lib.ffv2amd_mkv_write_packet(self._h, packet, len(packet), pts), "ffv2amd_mkv_write_packet")
        self._n = pts + 1

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            _lib.check(self._lib.ffv2amd_mkv_close(h), "ffv2amd_mkv_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
