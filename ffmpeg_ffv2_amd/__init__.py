"""MI355X-native FFV2 encode hot path.  Host mirror of the
reference's AVCodec init/encode2/close over a C-ABI + hand-written HIP kernels."""
from .encoder import FFV2Encoder, PIX_FMTS  # noqa: F401
from ._lib import FFV2Error  # noqa: F401
