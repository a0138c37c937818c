"""Host-side mirror of the reference encoder's interface for the FFV2 hot path.

Names follow reference libavcodec/ffv2enc.c: `FFV2Encoder(...)` is `ffv2enc_init`
(:495), `encode2()` is `ffv2_encode_frame` (:453), `close()` is `ffv2enc_close`
(:515).  Everything goes through the C-ABI in include/ffv2_amd.h; torch is used
only to own device memory and streams.
"""
import ctypes as C

import numpy as np

from . import _lib

PIX_FMTS = {  # allowed_pix_fmts, ffv2enc.c:596-601 (AVPixelFormat values of the reference tree)
    "gray": 8, "yuv444p": 5, "yuv444p10le": 70, "yuv444p12le": 133,
    "gbrp": 73, "gbrp10le": 77, "gbrp12le": 137,
}


class FFV2Encoder:
    def __init__(self, width, height, pix_fmt="yuv444p", device=0, max_batch=1):
        self._lib = _lib.load()
        if isinstance(pix_fmt, str):
            if pix_fmt not in PIX_FMTS:
                # avcodec_open2 refuses pix_fmts outside the whitelist (utils.c:814-822)
                raise _lib.FFV2Error(-22, "pix_fmt %s" % pix_fmt)
            pix_fmt = PIX_FMTS[pix_fmt]
        h = C.c_void_p()
        _lib.check(self._lib.ffv2amd_encoder_create(C.byref(h), width, height, pix_fmt, device, max_batch),
                   "ffv2amd_encoder_create")
        self._h = h
        self.info = _lib.Info()
        _lib.check(self._lib.ffv2amd_encoder_info(self._h, C.byref(self.info)), "ffv2amd_encoder_info")
        self.device = device
        self.dtype = np.dtype(np.uint8) if self.info.depth == 8 else np.dtype("<u2")

    # -- AVCodec.close --
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ffv2amd_encoder_destroy(self._h)
            self._h = None
            for ptr in getattr(self, "_lc_ptrs", []):        # lanecoder_finish's page-locked packet buffers
                self._lib.ffv2amd_host_free(ptr)
            self._lc_ptrs, self._lc_bufs = [], {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- AVCodec.encode2: host frame (P,H,W) -> packet bytes --
    def encode2(self, frame, qp=0, W=None):
        i = self.info
        frame = np.ascontiguousarray(frame, self.dtype)
        assert frame.shape == (i.planes, i.height, i.width), frame.shape
        data = (C.c_void_p * 4)()
        ls = (C.c_ssize_t * 4)()
        for p in range(i.planes):
            data[p] = frame[p].ctypes.data
            ls[p] = frame[p].strides[0]
        out = np.empty(i.packet_cap if qp == 0 else i.packet_cap_qp, np.uint8)
        n = C.c_size_t(0)
        wp = None
        if W is not None:
            W = np.ascontiguousarray(W, np.int32)
            assert W.size == i.block_planes
            wp = W.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.ffv2amd_encode_frame(self._h, data, ls, qp, wp,
                                                  out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)),
                   "ffv2amd_encode_frame")
        return out[:n.value].tobytes()

    # -- device-resident helpers --
    def pack_frames(self, frames):
        """(F,P,H,W) host array -> (F, frame_stride) uint8 host array in the device layout."""
        i = self.info
        frames = np.ascontiguousarray(frames, self.dtype)
        F = frames.shape[0]
        assert frames.shape[1:] == (i.planes, i.height, i.width)
        buf = np.zeros((F, i.frame_stride), np.uint8)
        bps = self.dtype.itemsize
        raw = frames.view(np.uint8).reshape(F, i.planes, i.height, i.width * bps)
        for p in range(i.planes):
            v = buf[:, p * i.plane_stride: p * i.plane_stride + i.row_pitch * i.height]
            v = v.reshape(F, i.height, i.row_pitch)
            v[:, :, : i.width * bps] = raw[:, p]
        return buf

    def upload(self, frames):
        import torch
        return torch.from_numpy(self.pack_frames(frames)).to("cuda:%d" % self.device)

    @staticmethod
    def _producer_done(*tensors):
        """The entry points without a stream argument run on the library's own (non-blocking) streams, which wait for
        nobody: whatever torch still has queued that writes these tensors has to be through first."""
        import torch
        for t in tensors:
            if t is not None:
                torch.cuda.current_stream(t.device).synchronize()

    def tstage(self, d_frames, want_coef=True, want_energy=True):
        """d_frames: torch uint8 (F, frame_stride) on this device -> (coef, energy) torch tensors."""
        import torch
        F = d_frames.shape[0]
        dev = d_frames.device
        coef = torch.empty((F, self.info.block_planes, 4096), dtype=torch.int32, device=dev) if want_coef else None
        en = torch.empty((F, self.info.block_planes, 13), dtype=torch.int64, device=dev) if want_energy else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._lib.ffv2amd_tstage_device(self._h, F, d_frames.data_ptr(),
                                                   coef.data_ptr() if want_coef else None,
                                                   en.data_ptr() if want_energy else None,
                                                   C.c_void_p(stream)), "ffv2amd_tstage_device")
        return coef, en

    def decode(self, packet, grid=False):
        """Decoder-side check (ffv2dec.c:315-377): packet -> ((P,H,W) samples, qp).  grid: with the reference
        decoder's DEBUGGING overwrite of every superblock's first row and column."""
        i = self.info
        out = np.zeros((i.planes, i.height, i.width), self.dtype)
        data = (C.c_void_p * 4)()
        ls = (C.c_ssize_t * 4)()
        for p in range(i.planes):
            data[p] = out[p].ctypes.data
            ls[p] = out[p].strides[0]
        buf = np.frombuffer(bytes(packet), np.uint8)
        qp = C.c_int(-1)
        _lib.check(self._lib.ffv2amd_decode_frame(self._h, buf.ctypes.data_as(C.c_void_p), buf.size, data, ls,
                                                  1 if grid else 0, C.byref(qp)), "ffv2amd_decode_frame")
        return out, qp.value

    def tstage_wide(self, d_frame):
        """One frame (torch uint8 (frame_stride,) or (1, frame_stride)) through the plain-int32 T-stage
        (ffv2_wide.hip): any 16-bit sample.  -> (coef (block_planes, 4096) int32, energy (block_planes, 13) int64)."""
        import torch
        dev = d_frame.device
        coef = torch.empty((self.info.block_planes, 4096), dtype=torch.int32, device=dev)
        en = torch.empty((self.info.block_planes, 13), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._lib.ffv2amd_tstage_wide_device(self._h, d_frame.data_ptr(), coef.data_ptr(), en.data_ptr(),
                                                        C.c_void_p(stream)), "ffv2amd_tstage_wide_device")
        return coef, en

    def alloc_packets(self, nframes):
        import torch
        dev = "cuda:%d" % self.device
        return (torch.empty((nframes, self.info.packet_cap), dtype=torch.uint8, device=dev),
                torch.empty((nframes,), dtype=torch.int32, device=dev),
                torch.empty((nframes,), dtype=torch.int32, device=dev))

    def encode_batch_device(self, d_frames, qp=0, d_W=None, out=None, stream=None):
        """Asynchronous: frames resident in HBM -> packets resident in HBM.
        Returns (packets (F,cap) uint8, sizes (F,) int32, status (F,) int32)."""
        import torch
        F = d_frames.shape[0]
        if out is None:
            out = self.alloc_packets(F)
        pk, sizes, status = out
        if stream is None:
            stream = torch.cuda.current_stream(d_frames.device).cuda_stream
        _lib.check(self._lib.ffv2amd_encode_batch_device(
            self._h, F, d_frames.data_ptr(), qp, d_W.data_ptr() if d_W is not None else None,
            pk.data_ptr(), pk.stride(0), sizes.data_ptr(), status.data_ptr(), C.c_void_p(stream)),
            "ffv2amd_encode_batch_device")
        return pk, sizes, status

    def encode_batch_to_host(self, d_frames, qp=0, d_W=None):
        """Synchronous batch encode2 for any qp >= 0: frames in HBM -> list of packets (bytes);
        a frame the reference would abort on raises FFV2Error(-1)."""
        F = d_frames.shape[0]
        cap = self.info.packet_cap if qp == 0 else self.info.packet_cap_qp
        pk = np.empty((F, cap), np.uint8)
        sizes = np.zeros(F, np.uint32)
        status = np.zeros(F, np.int32)
        self._producer_done(d_frames, d_W)
        _lib.check(self._lib.ffv2amd_encode_batch_to_host(
            self._h, F, d_frames.data_ptr(), qp, d_W.data_ptr() if d_W is not None else None,
            pk.ctypes.data_as(C.c_void_p), cap, sizes.ctypes.data_as(C.c_void_p),
            status.ctypes.data_as(C.c_void_p)), "ffv2amd_encode_batch_to_host")
        for f in range(F):
            if status[f] < 0:
                raise _lib.FFV2Error(int(status[f]), "frame %d" % f)
        return [pk[f, : sizes[f]].tobytes() for f in range(F)]

    # -- 4:2:0 front end (the reference tool's auto-inserted bicubic scale filter) --
    def _planes420(self, y, u, v):
        i = self.info
        cw, ch = (i.width + 1) // 2, (i.height + 1) // 2
        y = np.ascontiguousarray(y, self.dtype); u = np.ascontiguousarray(u, self.dtype); v = np.ascontiguousarray(v, self.dtype)
        assert y.shape == (i.height, i.width) and u.shape == (ch, cw) and v.shape == (ch, cw), (y.shape, u.shape)
        return y, u, v

    def encode2_420(self, y, u, v, qp=0):
        """yuv420p* host frame (Y (H,W); U, V (ceil(H/2), ceil(W/2))) -> packet bytes."""
        y, u, v = self._planes420(y, u, v)
        data = (C.c_void_p * 3)(y.ctypes.data, u.ctypes.data, v.ctypes.data)
        ls = (C.c_ssize_t * 3)(y.strides[0], u.strides[0], v.strides[0])
        out = np.empty(self.info.packet_cap if qp == 0 else self.info.packet_cap_qp, np.uint8)
        n = C.c_size_t(0)
        _lib.check(self._lib.ffv2amd_encode_frame_420(self._h, data, ls, qp, out.ctypes.data_as(C.c_void_p), out.size,
                                                      C.byref(n)), "ffv2amd_encode_frame_420")
        return out[: n.value].tobytes()

    def upconvert_420(self, y, u, v):
        """-> (3,H,W) yuv444p* samples as the GPU front end produces them."""
        import torch
        y, u, v = self._planes420(y, u, v)
        src = torch.from_numpy(np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).view(np.uint8)).to(
            "cuda:%d" % self.device)
        assert src.numel() == self._lib.ffv2amd_frame_bytes_420(self._h)
        dst = torch.zeros((1, self.info.frame_stride), dtype=torch.uint8, device=src.device)
        stream = torch.cuda.current_stream(src.device).cuda_stream
        _lib.check(self._lib.ffv2amd_upconvert_420_device(self._h, 1, src.data_ptr(), dst.data_ptr(), C.c_void_p(stream)),
                   "ffv2amd_upconvert_420_device")
        return self.unpack_frames(dst.cpu().numpy())[0]

    def set_device_coder(self, on=True):
        """qp > 0: run the adaptive range coder on the device (one wavefront per frame) instead of host threads."""
        _lib.check(self._lib.ffv2amd_encoder_set_device_coder(self._h, 1 if on else 0), "set_device_coder")

    # -- qp > 0 with many frames in flight (ffv2_lanecoder.hip) --
    def lanecoder_open(self, frames_in_flight, packet_cap=0, calls_in_flight=2, backs=None):
        """Size the device coder's HBM scratch (lanecoder_bytes_per_frame() per frame).  packet_cap: bytes
        reserved per packet (0 = info.packet_cap_qp); a frame that needs more comes back as NOSPACE.
        calls_in_flight: 2 to 4 submitted calls before a finish is due.  backs: range chains side by side
        (1..calls_in_flight; None = the library's default, one)."""
        if backs is None:
            _lib.check(self._lib.ffv2amd_lanecoder_open(self._h, int(frames_in_flight), int(packet_cap), int(calls_in_flight)),
                       "ffv2amd_lanecoder_open")
        else:
            _lib.check(self._lib.ffv2amd_lanecoder_open_ex(self._h, int(frames_in_flight), int(packet_cap), int(calls_in_flight),
                                                           int(backs)), "ffv2amd_lanecoder_open_ex")
        self._lc_cap = int(packet_cap) or self.info.packet_cap_qp

    def lanecoder_close(self):
        _lib.check(self._lib.ffv2amd_lanecoder_close(self._h), "ffv2amd_lanecoder_close")

    def lanecoder_bytes_per_frame(self, packet_cap=0, calls_in_flight=2, backs=None):
        if backs is None:
            return int(self._lib.ffv2amd_lanecoder_bytes_per_frame(self._h, int(packet_cap), int(calls_in_flight)))
        return int(self._lib.ffv2amd_lanecoder_bytes_per_frame_ex(self._h, int(packet_cap), int(calls_in_flight), int(backs)))

    def lanecoder_encode(self, d_frames, qp, d_W=None, packet_stride=None, as_arrays=False):
        """Up to frames_in_flight frames in HBM -> packets, the range coder running one frame per lane.
        as_arrays: return (packets[F, stride] uint8, sizes, status) without raising on a failed frame
        (the array is reused by the next call with the same shape)."""
        if getattr(self, "_lc_pending", None):
            raise _lib.FFV2Error(-22, "lanecoder_encode with submitted calls in flight")
        self._producer_done(d_frames, d_W)
        _lib.check(self._lib.ffv2amd_lanecoder_submit(self._h, d_frames.shape[0], d_frames.data_ptr(), qp,
                                                      d_W.data_ptr() if d_W is not None else None), "ffv2amd_lanecoder_submit")
        self._lc_pending = [(d_frames, d_W)]
        pk, sizes, status = self.lanecoder_finish(packet_stride)
        if as_arrays:
            return pk, sizes, status
        F = d_frames.shape[0]
        for f in range(F):
            if status[f] < 0:
                raise _lib.FFV2Error(int(status[f]), "frame %d" % f)
        return [pk[f, : sizes[f]].tobytes() for f in range(F)]

    def lanecoder_submit(self, d_frames, qp, d_W=None):
        """Asynchronous half of lanecoder_encode.  False when calls_in_flight calls are already in flight.  The
        frames must stay alive and untouched until the matching lanecoder_finish()."""
        self._producer_done(d_frames, d_W)
        r = self._lib.ffv2amd_lanecoder_submit(self._h, d_frames.shape[0], d_frames.data_ptr(), qp,
                                               d_W.data_ptr() if d_W is not None else None)
        if r == -11:
            return False
        _lib.check(r, "ffv2amd_lanecoder_submit")
        self._lc_pending = getattr(self, "_lc_pending", []) + [(d_frames, d_W)]
        return True

    def lanecoder_finish(self, packet_stride=None):
        """Packets of the oldest submitted call: (packets[F, stride] uint8, sizes, status)."""
        d_frames, _ = self._lc_pending.pop(0)
        F = d_frames.shape[0]
        cap = int(packet_stride or self.info.packet_cap_qp)
        bufs = getattr(self, "_lc_bufs", {})
        if (F, cap) not in bufs:
            # page-locked: the packets come back by DMA
            ptr = self._lib.ffv2amd_host_alloc(F * cap)
            if not ptr:
                raise MemoryError("ffv2amd_host_alloc(%d)" % (F * cap))
            self._lc_ptrs = getattr(self, "_lc_ptrs", []) + [ptr]
            bufs[(F, cap)] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(F * cap,)).reshape(F, cap)
            self._lc_bufs = bufs
        pk = bufs[(F, cap)]
        sizes = np.zeros(F, np.uint32)
        status = np.zeros(F, np.int32)
        _lib.check(self._lib.ffv2amd_lanecoder_finish(self._h, pk.ctypes.data_as(C.c_void_p), cap,
                                                      sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p)),
                   "ffv2amd_lanecoder_finish")
        return pk, sizes, status

    def lanecoder_stats(self):
        """(chain kernel ms, whole back ms, symbols of frame 0) of the call finished last."""
        c, b, n = C.c_float(0), C.c_float(0), C.c_uint32(0)
        _lib.check(self._lib.ffv2amd_lanecoder_stats(self._h, C.byref(c), C.byref(b), C.byref(n)), "ffv2amd_lanecoder_stats")
        return c.value, b.value, n.value

    def lanecoder_finish_packed(self):
        """Packets of the oldest submitted call as they lie on the device, in one copy:
        (buf uint8, offsets uint64, sizes, status); packet f = buf[offsets[f] : offsets[f] + sizes[f]].
        buf is page-locked and reused by the next call with the same number of frames."""
        d_frames, _ = self._lc_pending.pop(0)
        F = d_frames.shape[0]
        cap = F * ((getattr(self, "_lc_cap", 0) or self.info.packet_cap_qp) + 16)
        bufs = getattr(self, "_lc_bufs", {})
        if ("packed", cap) not in bufs:
            ptr = self._lib.ffv2amd_host_alloc(cap)
            if not ptr:
                raise MemoryError("ffv2amd_host_alloc(%d)" % cap)
            self._lc_ptrs = getattr(self, "_lc_ptrs", []) + [ptr]
            bufs[("packed", cap)] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(cap,))
            self._lc_bufs = bufs
        buf = bufs[("packed", cap)]
        offs = np.zeros(F, np.uint64)
        sizes = np.zeros(F, np.uint32)
        status = np.zeros(F, np.int32)
        _lib.check(self._lib.ffv2amd_lanecoder_finish_packed(
            self._h, buf.ctypes.data_as(C.c_void_p), cap, offs.ctypes.data_as(C.c_void_p),
            sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p)), "ffv2amd_lanecoder_finish_packed")
        return buf, offs, sizes, status

    def qp_submit(self, d_frames, qp, d_W=None):
        """GPU half of a qp > 0 batch (asynchronous).  False when two batches are already in flight."""
        self._producer_done(d_frames, d_W)
        r = self._lib.ffv2amd_qp_submit(self._h, d_frames.shape[0], d_frames.data_ptr(), qp,
                                        d_W.data_ptr() if d_W is not None else None)
        if r == -11:
            return False
        _lib.check(r, "ffv2amd_qp_submit")
        self._qp_frames = getattr(self, "_qp_frames", []) + [d_frames.shape[0]]
        return True

    def qp_finish(self):
        """Host half of the oldest submitted batch -> list of packets (bytes)."""
        F = self._qp_frames.pop(0)
        cap = self.info.packet_cap_qp
        if getattr(self, "_qp_out", None) is None or self._qp_out.shape[0] < F:
            self._qp_out = np.empty((F, cap), np.uint8)
        pk = self._qp_out
        sizes = np.zeros(F, np.uint32)
        status = np.zeros(F, np.int32)
        _lib.check(self._lib.ffv2amd_qp_finish(self._h, pk.ctypes.data_as(C.c_void_p), cap,
                                               sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p)),
                   "ffv2amd_qp_finish")
        for f in range(F):
            if status[f] < 0:
                raise _lib.FFV2Error(int(status[f]), "frame %d" % f)
        return [pk[f, : sizes[f]].tobytes() for f in range(F)]

    def inverse_tstage(self, d_coef):
        """Decoder-side inverse of the T-stage: torch int32 (F, block_planes, 4096) ->
        torch uint8 (F, frame_stride) in the device frame layout."""
        import torch
        F = d_coef.shape[0]
        out = torch.zeros((F, self.info.frame_stride), dtype=torch.uint8, device=d_coef.device)
        stream = torch.cuda.current_stream(d_coef.device).cuda_stream
        _lib.check(self._lib.ffv2amd_inverse_tstage_device(self._h, F, d_coef.data_ptr(), out.data_ptr(),
                                                           C.c_void_p(stream)), "inverse_tstage_device")
        return out

    def unpack_frames(self, buf):
        """Inverse of pack_frames: (F, frame_stride) uint8 host array -> (F,P,H,W) samples."""
        i = self.info
        buf = np.ascontiguousarray(buf, np.uint8)
        F = buf.shape[0]
        bps = self.dtype.itemsize
        out = np.empty((F, i.planes, i.height, i.width * bps), np.uint8)
        for p in range(i.planes):
            v = buf[:, p * i.plane_stride: p * i.plane_stride + i.row_pitch * i.height].reshape(F, i.height, i.row_pitch)
            out[:, p] = v[:, :, : i.width * bps]
        return out.view(self.dtype).reshape(F, i.planes, i.height, i.width)

    def pvq_search(self, X, K):
        """Device PVQ search on rows of X (count, N) float32 -> int16 pulses (count, N)."""
        import torch
        X = np.ascontiguousarray(X, np.float32)
        count, N = X.shape
        stride = N + 8
        Xp = np.zeros((count, stride), np.float32)
        Xp[:, :N] = X
        dX = torch.from_numpy(Xp).to("cuda:%d" % self.device)
        dy = torch.zeros((count, stride), dtype=torch.int16, device=dX.device)
        stream = torch.cuda.current_stream(dX.device).cuda_stream
        _lib.check(self._lib.ffv2amd_pvq_search_device(self._h, dX.data_ptr(), stride, N, K, count,
                                                       dy.data_ptr(), C.c_void_p(stream)), "pvq_search_device")
        return dy.cpu().numpy()[:, :N].astype(np.int32)

    def set_coef_sink(self, d_coef):
        """Also keep the coding-order coefficients of every batch encode in HBM
        (torch int32 (max_batch, block_planes, 4096)) or None to stop."""
        self._coef_sink = d_coef            # keep it alive
        _lib.check(self._lib.ffv2amd_encoder_set_coef_sink(
            self._h, d_coef.data_ptr() if d_coef is not None else None), "set_coef_sink")

    def set_pipelined(self, on=True):
        """E-stage of call n overlaps the T-stage of call n+1 (alternate two output sets)."""
        _lib.check(self._lib.ffv2amd_encoder_set_pipelined(self._h, 1 if on else 0), "set_pipelined")

    def flush(self, stream=None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream
        _lib.check(self._lib.ffv2amd_encoder_flush(self._h, C.c_void_p(stream)), "flush")

    # -- asynchronous frame ring: avcodec_send_frame / avcodec_receive_packet (encode.c:420,449) --
    def ring_open(self, depth=4):
        _lib.check(self._lib.ffv2amd_ring_open(self._h, depth), "ring_open")
        self._ring_out = np.empty(self.info.packet_cap, np.uint8)

    def ring_close(self):
        self._lib.ffv2amd_ring_close(self._h)

    def ring_pending(self):
        return self._lib.ffv2amd_ring_pending(self._h)

    def ring_send(self, frame, tag=0, W=None, pinned=False, register=False):
        """frame: (P,H,W) host array (any row stride).  Returns False when the ring is full
        (EAGAIN: receive a packet first).  pinned=True promises page-locked memory that stays
        untouched until the frame's packet has been received; register=True: ordinary memory from a pool of
        long-lived buffers, page-locked by the ring on first sight (FFV2AMD_FRAME_REGISTER)."""
        i = self.info
        assert frame.dtype == self.dtype and frame.shape == (i.planes, i.height, i.width), (frame.dtype, frame.shape)
        assert frame.strides[2] == self.dtype.itemsize
        data = (C.c_void_p * 4)()
        ls = (C.c_ssize_t * 4)()
        for p in range(i.planes):
            data[p] = frame[p].ctypes.data
            ls[p] = frame[p].strides[0]
        wp = None
        if W is not None:
            W = np.ascontiguousarray(W, np.int32)
            assert W.size == i.block_planes
            wp = W.ctypes.data_as(C.c_void_p)
        r = self._lib.ffv2amd_ring_send(self._h, data, ls, wp, int(tag), (1 if pinned else 0) | (4 if register else 0))
        if r == -11:
            return False
        _lib.check(r, "ring_send")
        return True

    def ring_send_420(self, y, u, v, tag=0, pinned=False, register=False):
        """A yuv420p* frame through the ring (Y (H,W); U, V (ceil(H/2), ceil(W/2)), any row stride): half the
        PCIe bytes of its 4:4:4 form, up-converted on the frame's compute stream.  False when the ring is full."""
        i = self.info
        cw, ch = (i.width + 1) // 2, (i.height + 1) // 2
        for a, shp in ((y, (i.height, i.width)), (u, (ch, cw)), (v, (ch, cw))):
            assert a.dtype == self.dtype and a.shape == shp and a.strides[1] == self.dtype.itemsize, (a.dtype, a.shape)
        data = (C.c_void_p * 3)(y.ctypes.data, u.ctypes.data, v.ctypes.data)
        ls = (C.c_ssize_t * 3)(y.strides[0], u.strides[0], v.strides[0])
        r = self._lib.ffv2amd_ring_send_420(self._h, data, ls, None, int(tag), (1 if pinned else 0) | (4 if register else 0))
        if r == -11:
            return False
        _lib.check(r, "ring_send_420")
        return True

    def pinned_frames_420(self, count):
        """count page-locked yuv420p* frames as a list of (Y, U, V) sample arrays whose row strides equal the
        device pitches (one DMA per plane), for ring_send_420(pinned=True).  Free with free_pinned()."""
        i = self.info
        bps = self.dtype.itemsize
        cw, ch = (i.width + 1) // 2, (i.height + 1) // 2
        cp = (cw * bps + 127) // 128 * 128
        per = i.row_pitch * i.height + 2 * cp * ch
        ptr = self._lib.ffv2amd_host_alloc(count * per)
        if not ptr:
            raise MemoryError("ffv2amd_host_alloc(%d)" % (count * per))
        self._pinned = getattr(self, "_pinned", []) + [ptr]
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * per,))
        out = []
        for n in range(count):
            b = buf[n * per: (n + 1) * per]
            yy = b[: i.row_pitch * i.height].reshape(i.height, i.row_pitch).view(self.dtype)[:, : i.width]
            uu = b[i.row_pitch * i.height:][: cp * ch].reshape(ch, cp).view(self.dtype)[:, : cw]
            vv = b[i.row_pitch * i.height + cp * ch:][: cp * ch].reshape(ch, cp).view(self.dtype)[:, : cw]
            out.append((yy, uu, vv))
        return out

    def ring_receive(self, wait=True):
        """-> (tag, packet bytes) of the oldest frame in flight, or None (nothing in flight /
        wait=False and not finished yet).  A failed frame raises FFV2Error."""
        n = C.c_size_t(0)
        tag = C.c_int64(0)
        r = self._lib.ffv2amd_ring_receive(self._h, self._ring_out.ctypes.data_as(C.c_void_p), self._ring_out.size,
                                           C.byref(n), C.byref(tag), 1 if wait else 0)
        if r == -11:
            return None
        _lib.check(r, "ring_receive")
        return tag.value, self._ring_out[: n.value].tobytes()

    # ---- send_frame / receive_packet at qp > 0 on top of the lane coder (ffv2amd_qpring_*) ----
    def qpring_open(self, qp, frames_per_call, packet_cap=0):
        _lib.check(self._lib.ffv2amd_qpring_open(self._h, int(qp), int(frames_per_call), int(packet_cap)), "qpring_open")
        cap = packet_cap or self.info.packet_cap_qp
        self._qpring_out = np.empty(int(cap) + 16, np.uint8)

    def qpring_send(self, frame, tag=0, W=None, pinned=False, yuv420=False, register=False):
        """frame: (P,H,W) host array, or with yuv420=True the (Y, U, V) arrays of a yuv420p* frame.  False: EAGAIN
        (receive packets first, then send the frame again)."""
        i = self.info
        planes = list(frame) if yuv420 else [frame[p] for p in range(i.planes)]
        data = (C.c_void_p * 4)()
        ls = (C.c_ssize_t * 4)()
        for p, a in enumerate(planes):
            assert a.dtype == self.dtype and a.strides[1] == self.dtype.itemsize, (a.dtype, a.strides)
            data[p] = a.ctypes.data
            ls[p] = a.strides[0]
        wp = None
        if W is not None:
            W = np.ascontiguousarray(W, np.int32)
            assert W.size == i.block_planes
            wp = W.ctypes.data_as(C.c_void_p)
        r = self._lib.ffv2amd_qpring_send(self._h, data, ls, wp, int(tag),
                                          (1 if pinned else 0) | (2 if yuv420 else 0) | (4 if register else 0))
        if r == -11:
            return False
        _lib.check(r, "qpring_send")
        return True

    def qpring_flush(self):
        r = self._lib.ffv2amd_qpring_flush(self._h)
        if r == -11:
            return False
        _lib.check(r, "qpring_flush")
        return True

    def qpring_receive(self, wait=True):
        """-> (tag, packet bytes), or None (EAGAIN).  A frame the reference would abort on raises FFV2Error(-1) and
        leaves the ring (its tag is in the exception's `tag`)."""
        n = C.c_size_t(0)
        tag = C.c_int64(0)
        r = self._lib.ffv2amd_qpring_receive(self._h, self._qpring_out.ctypes.data_as(C.c_void_p), self._qpring_out.size,
                                             C.byref(n), C.byref(tag), 1 if wait else 0)
        if r == -11:
            return None
        if r < 0:
            e = _lib.FFV2Error(r, "qpring_receive")
            e.tag = tag.value
            raise e
        return tag.value, self._qpring_out[: n.value].tobytes()

    def qpring_pending(self):
        return self._lib.ffv2amd_qpring_pending(self._h)

    def qpring_close(self):
        _lib.check(self._lib.ffv2amd_qpring_close(self._h), "qpring_close")

    def pinned_frames(self, count):
        """(count,P,H,W) sample array in page-locked host memory (row stride = the device row
        pitch), for ring_send(pinned=True).  Free with free_pinned()."""
        i = self.info
        nbytes = count * i.frame_stride
        ptr = self._lib.ffv2amd_host_alloc(nbytes)
        if not ptr:
            raise MemoryError("ffv2amd_host_alloc(%d)" % nbytes)
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
        planes = buf.reshape(count, i.planes, i.plane_stride)[:, :, : i.row_pitch * i.height]
        rows = planes.reshape(count, i.planes, i.height, i.row_pitch)
        arr = rows.view(self.dtype)[:, :, :, : i.width]
        self._pinned = getattr(self, "_pinned", []) + [ptr]
        return arr

    def free_pinned(self):
        for ptr in getattr(self, "_pinned", []):
            self._lib.ffv2amd_host_free(ptr)
        self._pinned = []

    def profile(self, on=True):
        """on: False/0 off, True/1 every call, n > 1 every n-th call (HIP timing events around the two kernels)."""
        _lib.check(self._lib.ffv2amd_profile_enable(self._h, int(on)), "profile_enable")

    def tstage_kernel_name(self, nframes):
        return self._lib.ffv2amd_tstage_kernel_name(self._h, int(nframes)).decode()

    def profile_read(self):
        """-> (tstage_ms_total, estage_ms_total, launches) since the last read."""
        t, x, n = C.c_double(0), C.c_double(0), C.c_int(0)
        _lib.check(self._lib.ffv2amd_profile_read(self._h, C.byref(t), C.byref(x), C.byref(n)), "profile_read")
        return t.value, x.value, n.value

    def profile_read_ex(self):
        """-> (tstage_ms_total, estage_ms_total, launches, shortest T-stage launch ms, longest) since the last read."""
        t, x, n, lo, hi = C.c_double(0), C.c_double(0), C.c_int(0), C.c_double(0), C.c_double(0)
        _lib.check(self._lib.ffv2amd_profile_read_ex(self._h, C.byref(t), C.byref(x), C.byref(n), C.byref(lo), C.byref(hi)),
                   "profile_read_ex")
        return t.value, x.value, n.value, lo.value, hi.value

    @staticmethod
    def collect(pk, sizes, status):
        """Synchronise and bring packets to the host as a list of bytes."""
        sizes = sizes.cpu().numpy()
        status = status.cpu().numpy()
        for f, st in enumerate(status):
            if st < 0:
                raise _lib.FFV2Error(int(st), "frame %d" % f)
        hp = pk.cpu().numpy()
        return [hp[f, : sizes[f]].tobytes() for f in range(hp.shape[0])]
