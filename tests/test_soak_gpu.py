"""Fixed-seed slices of the two soak tools, so that the random-geometry evidence is part of the suite
the driver runs (it lived in gpurun_out/ only).  tools/soak_parity.py: both T-stage
kernels, every accepted pix_fmt, 1-3 frames per launch, structured / noise / flat content, coefficients
+ energies + qp 0 packets vs the oracle, qp in {4, 16, 64} on some (aborts must agree), 4:2:0 through
the up-conversion kernel and the frame ring on some.  tools/soak_lanecoder.py: the many-frames-in-flight
coder against the host coder on frames of uneven length sharing a group of lanes.  About a minute."""
import pytest

pytestmark = pytest.mark.gpu


_PARITY_SLICE = """
import ctypes, json, os, sys
sys.path.insert(0, %r)
so = os.path.join(%r, "tools", "debug", "abort_trace.so")
if os.path.exists(so):
    ctypes.CDLL(so)                      # a dying process names who called abort(), and the case it was in
from tools import soak_parity
print(json.dumps(soak_parity.run(cases=1200, seed=20261004, max_w=700, max_h=500, p_qp=0.2, p_420=0.3, quiet=True)))
"""


def test_soak_parity_slice():
    """In a process of its own (FFV2_SOAK_INPROCESS=1: in this one).  Three of some thirty runs of the whole suite died
    of a SIGABRT inside this slice when it ran in the suite's process, at the first copy or synchronisation of a case
    (no message, no GPU fault, never in a process that runs the soak alone).  A process of
    its own is how the soak is meant to run (python tools/soak_parity.py), it cannot take the rest of the suite
    with it, and if it dies its stderr -- abort_trace.so's call stack included -- is the assertion's message."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("FFV2_SOAK_INPROCESS"):
        from tools import soak_parity
        n, nq, n420 = soak_parity.run(cases=1200, seed=20261004, max_w=700, max_h=500, p_qp=0.2, p_420=0.3, quiet=True)
    else:
        from ffmpeg_ffv2_amd import build
        build.build()                                  # the child finds the library built
        r = subprocess.run([sys.executable, "-c", _PARITY_SLICE % (root, root)], cwd=root, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, "soak_parity died with %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
        n, nq, n420 = json.loads(r.stdout.strip().splitlines()[-1])
    print("parity slice:", n, "geometries,", nq, "with qp > 0,", n420, "with 4:2:0")
    assert n == 1200 and nq >= 100 and n420 >= 60, (n, nq, n420)


def test_soak_lanecoder_slice():
    from tools import soak_lanecoder
    checked, aborted, nbytes = soak_lanecoder.run(cases=300, seed=20261004, max_frames=8, quiet=True)
    print("lanecoder slice:", checked, "frames identical,", aborted, "aborts agreed,", nbytes, "bytes")
    assert checked >= 800 and checked + aborted >= 1000, (checked, aborted)
