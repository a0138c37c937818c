"""Fixed-seed slices of the two soak tools, so that the random-geometry evidence is part of the suite
the driver runs (it lived in gpurun_out/ only).  tools/soak_parity.py: both T-stage
kernels, every accepted pix_fmt, 1-3 frames per launch, structured / noise / flat content, coefficients
+ energies + qp 0 packets vs the oracle, qp in {4, 16, 64} on some (aborts must agree), 4:2:0 through
the up-conversion kernel and the frame ring on some.  tools/soak_lanecoder.py: the many-frames-in-flight
coder against the host coder on frames of uneven length sharing a group of lanes.  About a minute."""
import pytest

pytestmark = pytest.mark.gpu


def test_soak_parity_slice():
    from tools import soak_parity
    n, nq, n420 = soak_parity.run(cases=1200, seed=20261004, max_w=700, max_h=500, p_qp=0.2, p_420=0.3, quiet=True)
    print("parity slice:", n, "geometries,", nq, "with qp > 0,", n420, "with 4:2:0")
    assert n == 1200 and nq >= 100 and n420 >= 60, (n, nq, n420)


def test_soak_lanecoder_slice():
    from tools import soak_lanecoder
    checked, aborted, nbytes = soak_lanecoder.run(cases=300, seed=20261004, max_frames=8, quiet=True)
    print("lanecoder slice:", checked, "frames identical,", aborted, "aborts agreed,", nbytes, "bytes")
    assert checked >= 800 and checked + aborted >= 1000, (checked, aborted)
