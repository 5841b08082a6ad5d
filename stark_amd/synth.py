"""Synthetic inputs: uniformly distributed canonical field elements, packed.

A counter-based SplitMix64 stream gives 128-bit candidates; candidates >= p are
rejected (about one in five), so the result is uniform on [0, p).  Vectorised
with numpy; the same (n, seed) always gives the same vector.
"""
from __future__ import annotations

import numpy as np

P = 407 * (1 << 119) + 1
_P_HI = np.uint64(P >> 64)
_P_LO = np.uint64(P & ((1 << 64) - 1))


def _splitmix64(counter: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = counter * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def field_vector(n: int, seed: int = 0x5741524B) -> np.ndarray:
    """(n, 2) uint64 array of packed elements [lo, hi], each < p."""
    out = np.empty((n, 2), dtype=np.uint64)
    filled = 0
    base = np.uint64((seed * 0x2545F4914F6CDD1D + 1) & ((1 << 64) - 1))
    pos = 0
    while filled < n:
        want = int((n - filled) * 1.3) + 64
        with np.errstate(over="ignore"):
            ctr = base + np.arange(2 * pos, 2 * (pos + want), dtype=np.uint64)
        r = _splitmix64(ctr).reshape(-1, 2)
        lo, hi = r[:, 0], r[:, 1]
        ok = (hi < _P_HI) | ((hi == _P_HI) & (lo < _P_LO))
        good = r[ok]
        k = min(len(good), n - filled)
        out[filled:filled + k] = good[:k]
        filled += k
        pos += want
    return out
