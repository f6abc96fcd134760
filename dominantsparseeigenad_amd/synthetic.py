"""Index-keyed synthetic vectors (counter-based, so any slab of a row-partitioned
vector can be produced independently on any rank and is identical everywhere).

element i of stream ``seed``  =  BoxMuller(u1, u2),  u1/u2 from splitmix64(seed, 2i) / (…, 2i+1)

Used for the Lanczos start vector q0, the CG start vector and the loss
direction t of the benchmark / parity workloads,
in place of the reference's ``torch.randn`` draws (reference Lanczos.py:52,59;
CG.py:58,121) which cannot be reproduced across devices.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform_bits(n, seed, offset=0):
    """uint64 stream: splitmix64 of (seed-mixed) counter ``offset .. offset+n``."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * _GOLDEN + np.uint64(0x1234567))
        return _splitmix64(ctr ^ key)


def normal_vector(n, seed, offset=0):
    """n standard normals (float64 numpy array) for global indices offset..offset+n."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * _GOLDEN + np.uint64(0x1234567))
        a = _splitmix64((ctr * np.uint64(2)) ^ key)
        b = _splitmix64((ctr * np.uint64(2) + np.uint64(1)) ^ key)
    # 53-bit mantissas -> (0,1]
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
    u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
