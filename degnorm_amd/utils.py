"""Host helpers shared by the single-node and sharded drivers."""
import numpy as np


def split_into_chunks(x, n):
    """
    Contiguous near-equal chunks with the reference's exact semantics (degnorm/utils.py:176-192):
    chunk size = ceil(len(x) / n), so fewer than n chunks may come back (len 201, n 20 -> 19 chunks).
    """
    size = int(np.ceil(len(x) / n))
    return [x[lo:lo + size] for lo in range(0, len(x), size)] if size > 0 else []


def partition_by_length(lengths, n):
    """
    Length-balanced gene partition for the sharded run: genes are dealt longest first to the n parts in boustrophedon
    order (0..n-1, n-1..0, ...), so every part gets the same number of genes (+-1) and nearly the same total length;
    inside a part the original gene order is kept.  The reference shards contiguous equal-count chunks
    (nmf_mpi.py:605); per-gene results do not depend on the partition, so the caller only has to
    un-permute the rows.  Always returns n lists (some empty when there are fewer genes than parts).
    """
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind='stable')
    parts = [[] for _ in range(int(n))]
    for k, g in enumerate(order):
        r = k % (2 * n)
        parts[r if r < n else 2 * n - 1 - r].append(int(g))
    return [sorted(q) for q in parts]


def chunk_bounds(length, n):
    """Start offsets (plus the end) of split_into_chunks(range(length), n)."""
    if length <= 0:
        return [0]
    size = int(np.ceil(length / n))
    return list(range(0, length, size)) + [length]
