"""Shared helpers for the test-suite (fixtures are flattened polyline lists: *_off, *_pts)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def unflat(g, name):
    off, pts = g[name + "_off"], g[name + "_pts"]
    return [pts[off[i]:off[i + 1]].reshape(-1, 1, 2) for i in range(len(off) - 1)]


def same_polys(a, b):
    if len(a) != len(b):
        return False
    return all(np.array_equal(np.asarray(x).reshape(-1, 2), np.asarray(y).reshape(-1, 2)) for x, y in zip(a, b))


def poly_multiset(polys):
    return sorted(tuple(np.asarray(p).reshape(-1).tolist()) for p in polys)
