"""Deterministic synthetic planar 4:4:4 frames.

S1 structured: v = (3x + 5y + 7n + 37p + ((x*y)>>5)) mod 2^depth
S2 noise     : numpy.random.default_rng(1234 + n).integers(0, 2^depth, (P,H,W))
"""
import numpy as np


def dtype_for(depth):
    return np.uint8 if depth == 8 else np.dtype("<u2")


def structured(n, planes, height, width, depth):
    y, x = np.mgrid[0:height, 0:width].astype(np.int64)
    out = np.empty((planes, height, width), dtype_for(depth))
    for p in range(planes):
        out[p] = ((3 * x + 5 * y + 7 * n + 37 * p + ((x * y) >> 5)) % (1 << depth)).astype(out.dtype)
    return out


def noise(n, planes, height, width, depth):
    return np.random.default_rng(1234 + n).integers(0, 1 << depth, (planes, height, width)).astype(dtype_for(depth))


def make(kind, n, planes, height, width, depth):
    return (structured if kind == "S1" else noise)(n, planes, height, width, depth)
