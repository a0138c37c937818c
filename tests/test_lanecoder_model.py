"""CPU check of the arithmetic behind ffv2_lanecoder.hip (tools/lanecoder_model.py): the range coder
split into CDF rows from prefix counts, the serial range chain, anchored 32-bit code words, the
final rounding as one more addend and a one-bit carry chain, against a direct transcription of the
coder (reference libavcodec/daala_entropy.c:107-151,328-379,428-440,624-735) on random streams."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lanecoder_model as M  # noqa: E402


def test_split_coder_equals_direct_coder_on_random_streams():
    M.self_check(rounds=120, seed=11)


def test_direct_coder_reproduces_a_survey_known_answer():
    """gray8 64x64 all 128 at qp 0 codes as 00 7f fe 18 -- header symbol (pix_fmt 8 >> 4 = 0
    of 13), one "no split" symbol, then raw bits only: 4 header bits, golomb(0), 4 tx bits, 14 x golomb(0)."""
    c = M.DirectCoder()
    c.encode(0, (32768 * 1 + 6) // 13, 32768)
    c.encode(0, 32 << 8, 128 << 8)                     # subdiv cdf {32,64,96,128} scaled to 15 bits
    bits = []

    def put(v, n):
        for i in range(n):
            bits.append((v >> i) & 1)
    put(8 & 15, 4); put(1, 1)                          # pix_fmt & 15, golomb(qp = 0)
    put(0, 4)                                          # tx type
    for _ in range(14):
        put(1, 1)                                      # "DC" and 13 gains, all zero
    c.rawbits = bits
    assert c.finish().hex() == "007ffe18"


def test_neutral_symbol_is_a_no_op_for_every_range():
    for rng in range(32768, 65536, 97):
        u, r = M.interval(rng, 0, 32768, 32768)
        assert (u, r) == (0, rng)


def test_kernel_form_of_the_interval_update_equals_the_reference_form():
    """lc_recur's algebra (d = min(t, x), g = sat(3 d - rng), packed 16-bit maps) against the interval
    update as daala_entropy.c:362-378 writes it, over the whole range of rng and random CDF triples
    with 16384 < ft <= 32768 (what `ft << sc` of daala_entropy.c:346 yields), including ft = 32768."""
    import random
    rnd = random.Random(5)
    for rng in list(range(32768, 65536, 61)) + [32768, 65535]:
        for _ in range(40):
            ft = rnd.choice([16385, 32768, rnd.randint(16385, 32768)])
            fh = rnd.randint(1, ft)
            fl = rnd.randint(0, fh - 1)
            assert M.interval_kernel(rng, fl, fh, ft) == M.interval(rng, fl, fh, ft), (rng, fl, fh, ft)
        assert M.interval_kernel(rng, 0, 32768, 32768) == (0, rng)
