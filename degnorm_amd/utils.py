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


def predicted_gene_cost(lengths, p=10, downsample_rate=1):
    """
    Relative cost of one outer iteration of a gene, from its length alone (known before anything is uploaded): the number of
    active columns (L, or ceil(L / rate) when down-sampling) plus the fixed part every inner iteration pays (reduction +
    eigen-solve), expressed in columns -- the constants are the measured per-column / per-iteration cycle counts of the
    class the gene will run in (~700-1 200 cycles per column per lane of 64-256 lanes, ~5 800 per inner
    iteration).  Only ratios matter.  Genes of the down-sampled regime cost the same fixed part each.
    """
    L = np.asarray(lengths, dtype=np.float64)
    if downsample_rate > 1:
        return np.ceil(L / float(downsample_rate)) + 64.0
    lanes = np.where(L > 4000, 256.0, np.where(L > 1888, 128.0, 64.0))       # wide / narrow / pair class (p = 10 boundaries)
    per_col = np.where(L > 4000, 1.25, 1.0)                                  # the wide class spills: ~25 % more per column
    # columns per lane x cost per column + fixed part worth ~6 columns per lane, times the SIMDs the gene occupies
    return (L / lanes * per_col + 6.0) * (lanes / 64.0)


def partition_by_cost(lengths, n, p=10, downsample_rate=1):
    """
    Gene partition for the sharded run balanced on PREDICTED COST (predicted_gene_cost) instead of on length: genes are
    taken most expensive first and each goes to the part with the least cost so far (LPT), ties to the part with fewer genes
    of the gene's class, so that every GPU also gets the same share of every gene class (each class is its own kernel and
    queue on a GPU).  Inside a part the original gene order is kept; always returns n lists.  Per-gene results do not depend
    on the partition; the reference shards contiguous equal-count chunks (nmf_mpi.py:605).
    """
    n = int(n)
    L = np.asarray(lengths, dtype=np.int64)
    cost = predicted_gene_cost(L, p, downsample_rate)
    cls = np.where(L > 4000, 0, np.where(L > 1888, 1, 2))
    order = np.lexsort((np.arange(len(L)), -cost))                           # most expensive first, stable
    parts = [[] for _ in range(n)]
    load = np.zeros(n)
    ncls = np.zeros((n, 3), dtype=np.int64)
    for g in order:
        c = cls[g]
        # least loaded part; among near-equal loads (within one gene's cost) the one with the fewest genes of this class
        lo = load.min()
        cand = np.flatnonzero(load <= lo + 0.5 * cost[g])
        r = int(cand[np.argmin(ncls[cand, c])])
        parts[r].append(int(g))
        load[r] += cost[g]
        ncls[r, c] += 1
    return [sorted(q) for q in parts]


def chunk_bounds(length, n):
    """Start offsets (plus the end) of split_into_chunks(range(length), n)."""
    if length <= 0:
        return [0]
    size = int(np.ceil(length / n))
    return list(range(0, length, size)) + [length]
