"""Inner products / scalar-vector products of the autograd glue, abstracted over WHERE a vector lives.

The backward formulas of the reference (symeig.py:80-84, CG.py:122,132-137) contain a handful of inner
products ``torch.matmul(u, v)`` and scalar-times-vector products.  On one device they are exactly those torch
expressions (``LOCAL``).  When the operator is row-partitioned over several GPUs (partitioned.py) a vector is
this rank's slab, an inner product must be closed by an all-reduce, and -- for second-order AD to stay
correct -- the derivative of ``s * v`` with respect to the replicated scalar ``s`` is again a GLOBAL inner
product.  ``space_of(A)`` returns the space the operand ``A`` lives in.
"""
from __future__ import annotations

import torch


class LocalSpace:
    """one device: the reference's own expressions"""

    partitioned = False

    @staticmethod
    def dot(a, b):
        return torch.matmul(a, b)

    @staticmethod
    def scale(s, v):
        return s * v


LOCAL = LocalSpace()


def space_of(A):
    """vector space of the operand: the operator object behind ``A`` (or behind its bound mat-vec method) may
    carry a ``space`` attribute (row-partitioned operators do); everything else lives on one device."""
    from . import engine

    owner = engine.native_of(A)
    sp = getattr(owner, "space", None) if owner is not None else None
    return sp if sp is not None else LOCAL
