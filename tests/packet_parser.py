"""Test-side restatement of the reference DECODER's entropy layer, used to check that the
packets our encoder writes decode back to the symbols that went in (encode -> decode round
trip of the bitstream).  Pure Python, small frames only.  Restates:
  libavcodec/daala_entropy.c:79-105   fillup / renormalize
                            :200-224  ff_daalaent_decode_bits   (raw bits, read from the packet end)
                            :273-326  daalaent_decode_cdf
                            :382-396  ff_daalaent_decode_uint
                            :413-425  ff_daalaent_decode_cdf_adapt
                            :564-578  ff_daalaent_decode_init
  libavcodec/ffv2dec.c:76-86 decode_golomb, :100-135 dequant_block (symbol order), :275-280 header
"""
M64 = (1 << 64) - 1
ABUNDANCE = 16384
BANDS_START = [0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4096]


def ilog(v):
    return v.bit_length()


class DaalaDec:
    def __init__(self, buf):
        self.b = bytes(buf)
        self.pos = 0                 # range bytes, read forward
        self.epos = len(self.b)      # raw bytes, read backward
        self.diff = 0
        self.rng = 0x8000
        self.cnt = -15
        self.win = 0
        self.nwin = 0
        self.raw_bits_read = 0
        self._fill()

    def _fill(self):
        i = 64 - 9 - (self.cnt + 15)
        while i >= 0 and self.pos < len(self.b):
            self.diff |= self.b[self.pos] << i
            self.cnt += 8
            i -= 8
            self.pos += 1
        if self.pos >= len(self.b):
            self.cnt = ABUNDANCE

    def _renorm(self, diff, rng):
        i = 16 - ilog(rng)
        self.diff = (diff << i) & M64
        self.rng = rng << i
        self.cnt -= i
        if self.cnt < 0:
            self._fill()

    def _decode(self, cdf, n, scale_mode):
        rng, diff = self.rng, self.diff
        cval = diff >> 48
        assert cval < rng, "corrupt stream"
        if scale_mode == "unscaled":
            ft = cdf[n - 1]
            assert 2 <= ft <= 32768
            scale = 15 - ilog(ft - 1)
            ft <<= scale
            assert ft <= rng
            if rng - ft >= ft:
                ft <<= 1
                scale += 1
            d = rng - ft
        else:                        # Q15
            assert cdf[n - 1] == 32768 and rng >= 32768
            d = rng - 32768
            ft = 32768
            scale = 0
        g = max(2 * d - ft, 0)
        # C: (2*cval + 1 - g)/3 truncates toward zero
        q = 2 * cval + 1 - g
        q = q // 3 if q >= 0 else -((-q) // 3)
        lim = max(max(cval >> 1, cval - d), q) >> scale
        ret, u = 0, 0
        v = cdf[0]
        while v <= lim:
            u = v
            ret += 1
            v = cdf[ret]
        u <<= scale
        v <<= scale
        sat = lambda a, b: a - min(a, b)
        u = u + min(u, g) + min(sat(u, g) >> 1, d)
        v = v + min(v, g) + min(sat(v, g) >> 1, d)
        self._renorm((diff - (u << 48)) & M64, v - u)
        return ret

    def bits(self, n):
        if self.nwin < n:
            while True:
                if self.epos <= 0:
                    self.nwin = ABUNDANCE
                    break
                self.epos -= 1
                self.win |= self.b[self.epos] << self.nwin
                self.nwin += 8
                if self.nwin > 64 - 8:
                    break
        r = self.win & ((1 << n) - 1)
        self.win >>= n
        self.nwin -= n
        self.raw_bits_read += n
        return r

    def uint(self, num):
        assert num > 16
        num -= 1
        bit = ilog(num) - 4
        adr = (num >> bit) + 1
        cdf = [(32768 * (k + 1) + adr // 2) // adr for k in range(adr)]   # daalatab.c uniform rows
        t = self._decode(cdf, adr, "q15")
        return (t << bit) | self.bits(bit)

    def adapt(self, cdf, n, inc):
        r = self._decode(cdf, n, "unscaled")
        if cdf[n - 1] + inc > 32767:
            for i in range(n):
                cdf[i] = (cdf[i] >> 1) + i + 1
        for i in range(r, n):
            cdf[i] += inc
        return r

    def golomb(self):
        c = 1
        while not self.bits(1):
            c = (c << 1) | self.bits(1)
        return c - 1


def parse_packet(pkt, nsb, planes):
    """-> dict(pix_fmt, qp, blocks=[(c0, [13 gains], [13 lists of signed pulses])])"""
    d = DaalaDec(pkt)
    pix_fmt = d.uint(196)
    qp = d.golomb()
    subdiv = [32, 64, 96, 128]
    test = [[j + 1 for j in range(qp)] for _ in range(13)]
    blocks = []
    for sb in range(nsb):
        split = d.adapt(subdiv, 4, 128)
        assert split == 0
        tx = d.bits(4)
        assert tx == 0
        for p in range(planes):
            c0 = d.golomb()
            if c0:
                c0 *= 1 - 2 * d.bits(1)
            gains, pulses = [], []
            for b in range(13):
                gains.append(d.golomb())
                ln = BANDS_START[b + 1] - BANDS_START[b]
                pc, band = 0, []
                for j in range(ln):
                    if pc >= qp:
                        break
                    q = d.adapt(test[b], qp, 64)
                    if q:
                        q *= 1 - 2 * d.bits(1)
                    band.append(q)
                    pc += abs(q)
                pulses.append(band)
            blocks.append((c0, gains, pulses))
    return {"pix_fmt": pix_fmt, "qp": qp, "blocks": blocks, "raw_bits": d.raw_bits_read}
