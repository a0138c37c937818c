"""ctypes mirrors of include/ffv2_amd_codec.h (the AVCodec-shaped shim) for the tests."""
import ctypes as C

MAX_DEVICES = 16
FRAME_PINNED, FRAME_YUV420 = 1, 2


class Ctx(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pix_fmt", C.c_int),
                ("global_quality", C.c_int), ("hip_device", C.c_int), ("ring_depth", C.c_int),
                ("priv_data", C.c_void_p), ("nb_devices", C.c_int), ("hip_devices", C.c_int * MAX_DEVICES),
                ("qp_frames_per_call", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("data", C.c_void_p * 4), ("linesize", C.c_ssize_t * 4), ("pts", C.c_int64)]


class Packet(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_int), ("pts", C.c_int64), ("dts", C.c_int64)]


def make_ctx(width, height, pix_fmt, qp=0, device=0, ring_depth=0, devices=None, qp_frames_per_call=0):
    ctx = Ctx(width, height, pix_fmt, qp, device, ring_depth, None)
    ctx.qp_frames_per_call = qp_frames_per_call
    if devices:
        ctx.nb_devices = len(devices)
        for i, d in enumerate(devices):
            ctx.hip_devices[i] = d
    return ctx


def frame_of(planes, pts):
    """planes: sequence of 2-D sample arrays (kept alive by the caller)."""
    f = Frame()
    for p, a in enumerate(planes):
        f.data[p] = a.ctypes.data
        f.linesize[p] = a.strides[0]
    f.pts = pts
    return f
