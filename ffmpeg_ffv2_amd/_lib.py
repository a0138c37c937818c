"""ctypes binding of libffv2amd.so (include/ffv2_amd.h).  No fallback: if the
shared library is missing this raises, it never routes to a CPU path."""
import ctypes as C
import os

# More hardware queues than the runtime's default of 4: the ring uses five streams, the lane coder four, and
# streams that share a queue wait for each other's kernels (ffv2_capi.cpp, ffv2amd_encoder_create).  Read by
# the HIP runtime when it initialises, so it has to be in the environment before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

PKG = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(PKG, "libffv2amd.so")

EXPORTS = [
    "ffv2amd_version", "ffv2amd_encoder_create", "ffv2amd_encoder_destroy", "ffv2amd_encoder_info",
    "ffv2amd_encode_frame", "ffv2amd_encode_batch_device", "ffv2amd_tstage_device",
    "ffv2amd_coded_gain", "ffv2amd_range_prefix",
    "ffv2amd_encoder_set_coef_sink", "ffv2amd_profile_enable", "ffv2amd_profile_read", "ffv2amd_profile_read_ex", "ffv2amd_tstage_kernel_name", "ffv2amd_debug_force_tstage",
    "ffv2amd_encoder_set_pipelined", "ffv2amd_encoder_flush", "ffv2amd_encode_batch_to_host", "ffv2amd_pvq_search_device", "ffv2amd_inverse_tstage_device",
    "ffv2amd_ring_open", "ffv2amd_ring_send", "ffv2amd_ring_receive", "ffv2amd_ring_pending", "ffv2amd_ring_close",
    "ffv2amd_host_alloc", "ffv2amd_host_free", "ffv2amd_qp_submit", "ffv2amd_qp_finish",
    "ffv2amd_ring_send_420", "ffv2amd_tstage_wide_device", "ffv2amd_decode_frame", "ffv2amd_parse_packet", "ffv2amd_qp_send_frame", "ffv2amd_qp_send_frame_420", "ffv2amd_qp_receive_packet", "ffv2amd_qp_pending", "ffv2amd_encoder_set_device_coder",
    "ffv2amd_qpring_open", "ffv2amd_qpring_send", "ffv2amd_qpring_flush", "ffv2amd_qpring_receive", "ffv2amd_qpring_pending", "ffv2amd_qpring_close",
    "ffv2amd_frame_bytes_420", "ffv2amd_upconvert_420_device", "ffv2amd_encode_frame_420",
    "ffv2amd_lanecoder_open", "ffv2amd_lanecoder_open_ex", "ffv2amd_lanecoder_close", "ffv2amd_lanecoder_bytes_per_frame",
    "ffv2amd_lanecoder_bytes_per_frame_ex", "ffv2amd_lanecoder_encode",
    "ffv2amd_lanecoder_submit", "ffv2amd_lanecoder_finish", "ffv2amd_lanecoder_finish_packed", "ffv2amd_lanecoder_stats", "ffv2amd_debug_lanecoder_window", "ffv2amd_debug_pvq_time",
    # AVCodec-shaped host shim (ffv2enc_amd.c)
    "ffv2amd_codec_init", "ffv2amd_codec_encode2", "ffv2amd_codec_close", "ffv2amd_codec_descriptor",
    "ffv2amd_codec_send_frame", "ffv2amd_codec_receive_packet", "ffv2amd_packet_unref", "ffv2amd_codec_encode_yuv420",
    # Matroska wire step (ffv2mkv.c)
    "ffv2amd_mkv_open", "ffv2amd_mkv_write_packet", "ffv2amd_mkv_close",
]

ERRORS = {-11: "EAGAIN", -22: "EINVAL", -12: "ENOMEM", -5: "EIO (HIP device/runtime)", -28: "ENOSPC",
          -34: "ERANGE (sample exceeds bit depth / gain table)", -1: "reference would abort",
          -38: "ENOSYS (not supported)"}


class Info(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pix_fmt", C.c_int),
                ("planes", C.c_int), ("depth", C.c_int),
                ("num_sb_x", C.c_int), ("num_sb_y", C.c_int),
                ("block_planes", C.c_int), ("max_batch", C.c_int),
                ("packet_cap", C.c_size_t), ("packet_cap_qp", C.c_size_t), ("tstage_bytes_per_frame", C.c_size_t),
                ("row_pitch", C.c_size_t), ("plane_stride", C.c_size_t), ("frame_stride", C.c_size_t)]


class FFV2Error(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed: %d %s" % (what, code, ERRORS.get(code, "")))
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise RuntimeError("%s is missing: build it with `python -m ffmpeg_ffv2_amd.build` "
                           "(hipcc, gfx950). There is no CPU fallback." % SO)
    # torch wheels bundle their own libamdhip64.so (same SONAME, libamdhip64.so.7, as
    # /opt/rocm's).  Whichever HIP runtime is mapped first serves every later NEEDED
    # entry with that SONAME; if ours came first torch would map a second runtime and
    # one of the two would see no GPU.  So: torch (when present) goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO)
    lib.ffv2amd_version.restype = C.c_char_p
    lib.ffv2amd_encoder_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ffv2amd_encoder_destroy.argtypes = [C.c_void_p]
    lib.ffv2amd_encoder_destroy.restype = None
    lib.ffv2amd_encoder_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    lib.ffv2amd_encode_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.ffv2amd_encode_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_tstage_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_encoder_set_coef_sink.argtypes = [C.c_void_p, C.c_void_p]
    lib.ffv2amd_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.ffv2amd_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.ffv2amd_profile_read_ex.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ffv2amd_tstage_kernel_name.argtypes = [C.c_void_p, C.c_int]
    lib.ffv2amd_tstage_kernel_name.restype = C.c_char_p
    lib.ffv2amd_debug_force_tstage.argtypes = [C.c_int]
    lib.ffv2amd_debug_force_tstage.restype = None
    lib.ffv2amd_encoder_set_pipelined.argtypes = [C.c_void_p, C.c_int]
    lib.ffv2amd_encoder_flush.argtypes = [C.c_void_p, C.c_void_p]
    lib.ffv2amd_encode_batch_to_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.ffv2amd_inverse_tstage_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_pvq_search_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p]
    lib.ffv2amd_ring_open.argtypes = [C.c_void_p, C.c_int]
    lib.ffv2amd_ring_send.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_void_p,
                                      C.c_int64, C.c_uint]
    lib.ffv2amd_ring_receive.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_int64), C.c_int]
    lib.ffv2amd_parse_packet.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_decode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t),
                                         C.c_uint, C.POINTER(C.c_int)]
    lib.ffv2amd_tstage_wide_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_ring_send_420.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_void_p,
                                          C.c_int64, C.c_uint]
    lib.ffv2amd_qp_send_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_int, C.c_void_p, C.c_int64]
    lib.ffv2amd_qp_receive_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    lib.ffv2amd_qp_pending.argtypes = [C.c_void_p]
    lib.ffv2amd_qpring_open.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    lib.ffv2amd_qpring_send.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_void_p, C.c_int64, C.c_uint]
    lib.ffv2amd_qpring_flush.argtypes = [C.c_void_p]
    lib.ffv2amd_qpring_receive.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64), C.c_int]
    lib.ffv2amd_qpring_pending.argtypes = [C.c_void_p]
    lib.ffv2amd_qpring_close.argtypes = [C.c_void_p]
    lib.ffv2amd_ring_pending.argtypes = [C.c_void_p]
    lib.ffv2amd_ring_close.argtypes = [C.c_void_p]
    lib.ffv2amd_ring_close.restype = None
    lib.ffv2amd_host_alloc.argtypes = [C.c_size_t]
    lib.ffv2amd_host_alloc.restype = C.c_void_p
    lib.ffv2amd_host_free.argtypes = [C.c_void_p]
    lib.ffv2amd_host_free.restype = None
    lib.ffv2amd_encoder_set_device_coder.argtypes = [C.c_void_p, C.c_int]
    lib.ffv2amd_qp_submit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.ffv2amd_qp_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.ffv2amd_lanecoder_open.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int]
    lib.ffv2amd_lanecoder_close.argtypes = [C.c_void_p]
    lib.ffv2amd_lanecoder_bytes_per_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lib.ffv2amd_lanecoder_bytes_per_frame.restype = C.c_size_t
    lib.ffv2amd_lanecoder_open_ex.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int]
    lib.ffv2amd_lanecoder_bytes_per_frame_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    lib.ffv2amd_lanecoder_bytes_per_frame_ex.restype = C.c_size_t
    lib.ffv2amd_lanecoder_submit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.ffv2amd_lanecoder_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.ffv2amd_lanecoder_finish_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_lanecoder_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    lib.ffv2amd_debug_pvq_time.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.ffv2amd_debug_lanecoder_window.argtypes = [C.c_uint32]
    lib.ffv2amd_debug_lanecoder_window.restype = None
    lib.ffv2amd_lanecoder_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.ffv2amd_frame_bytes_420.argtypes = [C.c_void_p]
    lib.ffv2amd_frame_bytes_420.restype = C.c_size_t
    lib.ffv2amd_upconvert_420_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ffv2amd_encode_frame_420.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.c_int,
                                             C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.ffv2amd_coded_gain.argtypes = [C.c_int64]
    lib.ffv2amd_coded_gain.restype = C.c_uint32
    lib.ffv2amd_range_prefix.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    _lib = lib
    return lib


def check(code, what):
    if code < 0:
        raise FFV2Error(code, what)
    return code
