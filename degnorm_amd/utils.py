"""Host helpers shared by the single-node and sharded drivers."""
import numpy as np


def split_into_chunks(x, n):
    """
    Contiguous near-equal chunks with the reference's exact semantics (degnorm/utils.py:176-192):
    chunk size = ceil(len(x) / n), so fewer than n chunks may come back (len 201, n 20 -> 19 chunks).
    """
    size = int(np.ceil(len(x) / n))
    return [x[lo:lo + size] for lo in range(0, len(x), size)] if size > 0 else []


def chunk_bounds(length, n):
    """Start offsets (plus the end) of split_into_chunks(range(length), n)."""
    if length <= 0:
        return [0]
    size = int(np.ceil(length / n))
    return list(range(0, length, size)) + [length]
