"""Encode -> decode round trip of the bitstream: packets written by the (pinned) oracle encoder are
read back with a restatement of the reference decoder's entropy layer (tests/packet_parser.py) and
must yield the symbols that went in.  This pins the adaptive range coder (qp > 0) from the decoder
side as well, independently of the encoder restatements.  CPU only, small frames."""
import numpy as np
import pytest

from ffmpeg_ffv2_amd import frames as synth
from tests.packet_parser import parse_packet, BANDS_START
from tests.oracle_lib import PIX


@pytest.mark.parametrize("fmt,P,H,W,depth", [("gray", 1, 64, 64, 8), ("yuv444p", 3, 100, 150, 8),
                                             ("yuv444p10le", 3, 70, 130, 10), ("gbrp12le", 3, 64, 64, 12)])
@pytest.mark.parametrize("qp", [0, 4, 16])
def test_packets_decode_back_to_their_symbols(oracle, fmt, P, H, W, depth, qp):
    fr = synth.noise(3, P, H, W, depth)
    pkt = oracle.encode(fr, fmt, qp=qp)
    coef, en = oracle.tstage(fr, fmt)
    nsb = ((W + 63) // 64) * ((H + 63) // 64)
    out = parse_packet(pkt, nsb, P)
    assert out["pix_fmt"] == PIX[fmt] and out["qp"] == qp
    assert len(out["blocks"]) == nsb * P
    for bp, (c0, gains, pulses) in enumerate(out["blocks"]):
        assert c0 == coef[bp, 0]
        assert gains == [oracle.coded_gain(int(e)) for e in en[bp]]
        for b in range(13):
            lo, ln = 1 + BANDS_START[b], BANDS_START[b + 1] - BANDS_START[b]
            if qp == 0:
                assert pulses[b] == []
                continue
            x = np.zeros(ln, np.float32)
            n = min(ln, 4096 - lo)
            g = np.float32(np.sqrt(np.float32(en[bp, b]))) + np.float32(np.finfo(np.float32).eps)
            x[:n] = coef[bp, lo: lo + n].astype(np.float32) / g
            y = oracle.pvq_search(x, qp)
            # the coder stops once qp pulses have been seen (ffv2enc.c:177)
            k, pc = 0, 0
            while k < ln and pc < qp:
                pc += abs(int(y[k]))
                k += 1
            assert pulses[b] == [int(v) for v in y[:k]], (bp, b)
    # every raw bit of the tail was consumed except the zero padding of the last raw byte
    assert out["raw_bits"] > 0
