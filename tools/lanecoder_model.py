#!/usr/bin/env python3
"""Arithmetic model of the many-frames-in-flight range coder (ffv2_lanecoder.hip), in plain Python.

The Daala range coder (reference libavcodec/daala_entropy.c:107-151,328-379,624-735) is one serial
chain per frame, but only the *range* recurrence is serial.  The model splits a frame's coder into
the pieces the kernels implement and checks them against a direct transcription of the coder:

  cdf rows    adaptive CDF rows advance with the symbols alone (:428-440), never with the range:
              between two halvings a row is its value at the last halving plus 64 x a prefix
              count, so (fl, fh, ft) of every symbol come from prefix counts over chunks.
  chain       rng' = f(rng, fl, fh, ft): the only serial part; it yields per symbol the offset u
              added to `low` and the shift d.  `low` itself is never carried along: the final
              code is  sum_k u_k << (T - D_k),  D_k = shifts before symbol k, T = all shifts.
  words       that sum is accumulated into 32-bit words anchored every 16 bits of depth (a symbol
              shifts by at most 15, so the anchor advances by 0 or 1 per symbol), then turned into
              bytes with a one-bit carry chain.
  done        ff_daalaent_encode_done's rounding (:624-674) is one more addend at depth T.

usage: python tools/lanecoder_model.py   (self-check on random symbol streams)
"""
import random


def ilog(v):
    return v.bit_length()


def interval(rng, fl, fh, ft):
    """daala_entropy.c:362-378 for 16384 < ft <= 32768 <= rng < 65536 -> (u, r)"""
    sc = 1 if (rng - ft) >= ft else 0
    fl <<= sc; fh <<= sc; ft <<= sc
    d = rng - ft
    g = max(2 * d - ft, 0)

    def m(x):
        return x + min(x, g) + min(max(x - g, 0) >> 1, d)
    u, v = m(fl), m(fh)
    return u, v - u


def interval_kernel(rng, fl, fh, ft):
    """the same interval as lc_recur computes it (ffv2_lanecoder.hip): unsigned 32-bit min instead of
    the compare, g from 3 d - rng, both bounds in 16-bit halves of one word"""
    M = 0xFFFFFFFF
    t = (rng - ft) & M
    x = (t - ft) & M
    d = min(t, x)
    g = max(3 * d - rng, 0)
    sc = 1 - (x >> 31)
    lo = ((fl | (fh << 16)) << sc) & M
    out = []
    for h in (lo & 0xFFFF, lo >> 16):
        b = max(h - g, 0) >> 1
        v = h + min(h, g) + min(b, d)
        assert v < 65536 and g < 65536 and d < 65536, "the packed 16-bit lanes must not overflow"
        out.append(v)
    return out[0], out[1] - out[0]


class DirectCoder:
    """the coder as the reference runs it: 64-bit window, 16-bit pre-carry words"""

    def __init__(self):
        self.low, self.rng, self.cnt, self.pre = 0, 0x8000, -9, []
        self.rawbits = []                      # raw bits in write order

    def encode(self, fl, fh, ft):
        u, r = interval(self.rng, fl, fh, ft)
        l = self.low + u
        d = 16 - ilog(r)
        c, s = self.cnt, self.cnt + d
        if s >= 0:
            c += 16
            m = (1 << c) - 1
            if s >= 8:
                self.pre.append(l >> c); l &= m; c -= 8; m >>= 8
            self.pre.append(l >> c)
            s = c + d - 24
            l &= m
        self.low, self.rng, self.cnt = l << d, r << d, s

    def bits(self, v, n):
        for i in range(n):
            self.rawbits.append((v >> i) & 1)

    def finish(self):
        low, rng = self.low, self.rng
        m = 0x7FFF; e = (low + m) & ~m; s = 9; c = self.cnt
        while (e | m) >= low + rng:
            s += 1; m >>= 1; e = (low + m) & ~m
        s += c
        if s > 0:
            n = (1 << (c + 16)) - 1
            while True:
                self.pre.append(e >> (c + 16)); e &= n; s -= 8; c -= 8; n >>= 8
                if s <= 0:
                    break
        head = [0] * len(self.pre)
        carry = 0
        for i in range(len(self.pre) - 1, -1, -1):
            carry += self.pre[i]; head[i] = carry & 255; carry >>= 8
        return assemble(head, -s, self.rawbits)


def assemble(head, slack, rawbits):
    """daala_entropy.c:676-721: raw bytes behind the range bytes in reverse write order, the
    youngest <= slack raw bits OR-ed into the last range byte"""
    R = len(rawbits)
    nraw = max(0, -(-(R - slack) // 8))
    val = lambda j: sum(rawbits[8 * j + i] << i for i in range(8) if 8 * j + i < R)
    out = head + [0] * nraw
    for j in range(nraw):
        out[len(out) - 1 - j] = val(j)
    if R - 8 * nraw > 0:
        assert head, "the reference asserts here"
        out[len(head) - 1] |= val(nraw)
    return bytes(out)


def chain(symbols):
    """the serial part as the kernel runs it: state = (rng, o, acc, woff); the depth of the next
    symbol is 16 woff + 1 - o, its offset lands at bit o of word woff.  Returns the words and the
    final state."""
    rng, o, acc, woff = 0x8000, 1, 0, 0
    W = {}
    for fl, fh, ft in symbols:
        u, r = interval(rng, fl, fh, ft)
        d = 16 - ilog(r)
        rng = r << d
        acc += u << o
        assert acc < 1 << 32
        W[woff] = acc                      # stored every time; the last store of a word stands
        o -= d
        if o < 0:
            o += 16; woff += 1; acc = 0
    W[woff] = acc
    W[woff + 1] = 0
    return W, woff, o, rng


def finish_words(W, woff, o, rng, rawbits):
    T = 16 * woff + 1 - o
    npre = (T - 1) >> 3 if T >= 1 else 0
    cnt = -9 + T - 8 * npre
    # the coder's window: the low cnt + 24 bits of the code so far, read back from the last words
    V = W[woff] + (W.get(woff - 1, 0) << 16) + (W.get(woff - 2, 0) << 32)
    low = (V >> o) & ((1 << (cnt + 24)) - 1)
    m = 0x7FFF; e = (low + m) & ~m; s = 9
    while (e | m) >= low + rng:
        s += 1; m >>= 1; e = (low + m) & ~m
    s += cnt
    extra = (s + 7) >> 3 if s > 0 else 0
    slack = 8 * extra - s if s > 0 else -s
    nbytes = npre + extra
    W = dict(W)
    W[woff] += (e - low) << o
    assert W[woff] < 1 << 32
    word = lambda w: W.get(w, 0)
    # the sum is zero below the last byte only after the carries have run: start the chain at the
    # last word there is, keep the first nbytes
    top = 2 * woff + 2
    S = []
    for i in range(top):
        w = i >> 1
        if i & 1:
            S.append((word(w) & 255) + ((word(w + 1) >> 16) & 255))
        else:
            S.append(((word(w) >> 8) & 255) + ((word(w + 1) >> 24) & 255))
    body = [0] * top
    carry = 0
    for i in range(top - 1, -1, -1):
        carry += S[i]; body[i] = carry & 255; carry >>= 8
    assert carry == 0 and not any(body[nbytes:]), "the rounded code has nothing below its last byte"
    head = body[:nbytes]
    return assemble(head, slack, rawbits)


def cdf_rows_chunked(vals, n, inc=64, chunk=64):
    """(fl, fh, ft) << sc of every symbol of one CDF row from prefix counts over chunks that end
    at a halving (daala_entropy.c:428-440 restated)."""
    R = [i + 1 for i in range(n)]
    out, k = [], 0
    while k < len(vals):
        F0 = R[n - 1]
        th = 0 if F0 + inc > 32767 else -(-(32768 - inc - F0) // inc)
        m = min(chunk, len(vals) - k, th + 1)
        halve = m == th + 1
        xs = vals[k:k + m]
        for t, x in enumerate(xs):
            cl = sum(1 for y in xs[:t] if y <= x - 1)
            ch = sum(1 for y in xs[:t] if y <= x)
            fl = R[x - 1] + inc * cl if x else 0
            fh = R[x] + inc * ch
            ft = F0 + inc * t
            sc = 15 - ilog(ft - 1)
            out.append((fl << sc, fh << sc, ft << sc))
        last = xs[-1]
        for i in range(n):
            ca = sum(1 for y in xs if y <= i)
            if halve:
                R[i] = ((R[i] + inc * (ca - (1 if last <= i else 0))) >> 1) + i + 1 + (inc if last <= i else 0)
            else:
                R[i] += inc * ca
        k += m
    return out


def cdf_rows_direct(vals, n, inc=64):
    cdf = [i + 1 for i in range(n)]
    out = []
    for x in vals:
        fl, fh, ft = (cdf[x - 1] if x else 0), cdf[x], cdf[n - 1]
        sc = 15 - ilog(ft - 1)
        out.append((fl << sc, fh << sc, ft << sc))
        if cdf[n - 1] + inc > 32767:
            cdf = [(c >> 1) + i + 1 for i, c in enumerate(cdf)]
        cdf = [c + inc if i >= x else c for i, c in enumerate(cdf)]
    return out


def self_check(rounds=300, seed=7):
    rnd = random.Random(seed)
    for it in range(rounds):
        n = rnd.choice([2, 3, 16, 17, 40, 64])
        length = rnd.choice([0, 1, 2, 5, 100, 700, 3000])
        skew = rnd.choice([1, 3, 8])
        vals = [min(n - 1, int(rnd.random() ** skew * n)) for _ in range(length)]
        syms = cdf_rows_chunked(vals, n, chunk=rnd.choice([1, 7, 64]))
        assert syms == cdf_rows_direct(vals, n), "cdf rows"
        hdr = (0, 2521, 32768)
        syms = [hdr] + syms
        raw = [rnd.getrandbits(1) for _ in range(rnd.choice([0, 1, 5, 8, 9, 77]))]
        dc = DirectCoder()
        for s in syms:
            dc.encode(*s)
        dc.rawbits = list(raw)
        want = dc.finish()
        got = finish_words(*chain(syms), raw)
        assert got == want, (it, n, length, got.hex(), want.hex())
        # neutral symbols (probability one) change nothing
        W1 = chain(syms + [(0, 32768, 32768)] * 5)
        assert finish_words(*W1, raw) == want
    print("lane coder model: %d random streams agree with the direct coder" % rounds)


if __name__ == "__main__":
    self_check()
