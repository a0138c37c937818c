#!/usr/bin/env python3
"""LDS bank-conflict model of the T-stage's back half (phases D-F of ffv2_kernels.hip) under the gfx950
rules of MI355X_MICROARCH.md (LDS section): ds_read_b32 / ds_write_b32 are serviced in two groups of 32
lanes, bank = (byte address / 4) mod 32, an N-way conflict inside a group costs N cycles for that group.
Prints LDS-array cycles per block-plane for the transposition buffer pitch (XPITCH) and the raster buffer
pitch (RPITCH), and for XOR-swizzled raster layouts, so that the pitches the kernel ships with can be
held against every alternative.  usage: python tools/lds_bank_model.py"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan_lut():
    txt = open(os.path.join(ROOT, "ffmpeg_ffv2_amd", "csrc", "gen", "scan_lut.h")).read()
    start = txt.index("FFV2_SCAN_LUT[4096]")
    body = txt[txt.index("{", start) + 1: txt.index("}", start)]
    v = [int(x) for x in re.findall(r"\d+", body)]
    assert len(v) == 4096
    return v


def cycles(addrs):
    """addrs: 64 dword addresses (one per lane) of one ds_*_b32 instruction -> LDS-array cycles."""
    tot = 0
    for half in (addrs[:32], addrs[32:]):
        per_bank = {}
        for a in half:
            per_bank.setdefault(a % 32, set()).add(a)       # identical addresses broadcast
        tot += max(len(s) for s in per_bank.values())
    return tot


def gather_cycles(addr_of):
    """phase F: lane owns coding indices q = 256*(2i + e/4) + 4*lane + e%4; one ds_read_b32 per (i, e)."""
    lut = scan_lut()
    tot = 0
    for i in range(8):
        for e in range(8):
            a = []
            for lane in range(64):
                q = 256 * (2 * i + e // 4) + 4 * lane + (e & 3)
                r = lut[q]
                a.append(addr_of(r >> 6, r & 63))
            tot += cycles(a)
    return tot


def main():
    print("transposition buffer xb[lane*P + v] (write, v fixed) / xb[k*P + lane] (read): cycles per 64 instructions")
    for P in (64, 65, 66, 67, 68, 69, 70, 71, 72, 73):
        w = sum(cycles([lane * P + v for lane in range(64)]) for v in range(64))
        r = sum(cycles([k * P + lane for lane in range(64)]) for k in range(64))
        print("  XPITCH %2d: write %4d  read %4d   (conflict-free = 128 each)" % (P, w, r))
    print("raster buffer xb[lane*P + u] (write) and the scan-order gather (read): cycles per 64 instructions")
    best = None
    for P in range(64, 82):
        w = sum(cycles([lane * P + u for lane in range(64)]) for u in range(64))
        g = gather_cycles(lambda y, x: y * P + x)
        print("  RPITCH %2d: write %4d  gather %4d" % (P, w, g))
        if w == 128 and (best is None or g < best[1]):
            best = (P, g)
    print("  best linear pitch with conflict-free writes: %d (gather %d cycles, ideal 128)" % best)
    print("XOR-swizzled raster layouts  addr = y*64 + (x ^ f(y))  (writes stay conflict-free when f is a bijection per row)")
    for name, f in (("x ^ (y & 31)", lambda y: y & 31), ("x ^ ((y*5) & 31)", lambda y: (y * 5) & 31),
                    ("x ^ ((y >> 1) & 31)", lambda y: (y >> 1) & 31), ("x ^ ((y*3) & 63)", lambda y: (y * 3) & 63),
                    ("x ^ (y & 63)", lambda y: y & 63)):
        w = sum(cycles([lane * 64 + (u ^ f(lane)) for lane in range(64)]) for u in range(64))
        g = gather_cycles(lambda y, x: y * 64 + (x ^ f(y)))
        print("  %-22s write %4d  gather %4d" % (name, w, g))


if __name__ == "__main__":
    main()
