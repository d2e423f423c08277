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


P10_CLASS_LENGTHS = (4000, 1888)      # (split_len, tiny_len) of a p = 10 cohort on MI355X (dn_class_lengths): only a default for callers without a device


def gene_classes(lengths, split_len, tiny_len):
    """Class of every gene as the device will run it: 0 wide (longer than split_len), 1 narrow, 2 pair (at most tiny_len bases);
    split_len = 0: one class (include/degnorm_amd.h dn_class_lengths)."""
    L = np.asarray(lengths, dtype=np.int64)
    if not split_len or split_len <= 0:
        return np.zeros(len(L), dtype=np.int64)
    return np.where(L > split_len, 0, np.where((tiny_len or 0) > 0, np.where(L > tiny_len, 1, 2), 1))


def predicted_gene_cost(lengths, p=10, downsample_rate=1, class_lengths=None):
    """
    Relative cost of one outer iteration of a gene, from its length alone (known before anything is uploaded): the number of
    active columns (L, or ceil(L / rate) when down-sampling) plus the fixed part every inner iteration pays (reduction +
    eigen-solve), expressed in columns -- the constants are the measured per-column / per-iteration cycle counts of the
    class the gene will run in.  Only ratios matter.  Genes of the down-sampled regime cost the same fixed
    part each.  `class_lengths` = (split_len, tiny_len) of THIS cohort on the device (Device.class_lengths(p): the boundaries
    depend on p -- register and LDS capacity per column); without them p = 10 gets its known boundaries and any other sample
    count is treated as one class.
    """
    L = np.asarray(lengths, dtype=np.float64)
    if downsample_rate > 1:
        return np.ceil(L / float(downsample_rate)) + 64.0
    if class_lengths is None:
        class_lengths = P10_CLASS_LENGTHS if int(p) == 10 else (0, 0)
    cls = gene_classes(L, class_lengths[0], class_lengths[1])
    if not class_lengths[0]:
        return L / 256.0 + 4.0                                               # one class: columns per lane + the fixed part
    lanes = np.where(cls == 0, 256.0, np.where(cls == 1, 128.0, 64.0))        # wide / narrow / pair class
    per_col = np.where(cls == 0, 1.25, 1.0)                                   # the wide class spills: ~25 % more per column
    # columns per lane x cost per column + fixed part worth ~4 columns per lane (round 4: the vector-pipe eigen-solve; 6 before),
    # times the SIMDs the gene occupies
    return (L / lanes * per_col + 4.0) * (lanes / 64.0)


def partition_by_cost(lengths, n, p=10, downsample_rate=1, class_lengths=None):
    """
    Gene partition for the sharded run balanced on PREDICTED COST (predicted_gene_cost) instead of on length: genes are
    taken most expensive first and each goes to the part with the least cost so far (LPT), ties to the part with fewer genes
    of the gene's class, so that every GPU also gets the same share of every gene class (each class is its own kernel and
    queue on a GPU; `class_lengths`: see predicted_gene_cost).  Inside a part the original gene order is kept; always returns n
    lists.  Per-gene results do not depend on the partition; the reference shards contiguous equal-count chunks
    (nmf_mpi.py:605).
    """
    n = int(n)
    L = np.asarray(lengths, dtype=np.int64)
    if class_lengths is None:
        class_lengths = P10_CLASS_LENGTHS if int(p) == 10 else (0, 0)
    cost = predicted_gene_cost(L, p, downsample_rate, class_lengths)
    cls = gene_classes(L, class_lengths[0], class_lengths[1]) if downsample_rate <= 1 else np.zeros(len(L), dtype=np.int64)
    order = np.lexsort((np.arange(len(L)), -cost)).tolist()                  # most expensive first, stable
    parts = [[] for _ in range(n)]
    load = [0.0] * n                                                         # plain lists: n is small, 20 000 numpy calls are not
    ncls = [[0, 0, 0] for _ in range(n)]
    cost_l, cls_l = cost.tolist(), cls.tolist()
    for g in order:
        c, cg = cls_l[g], cost_l[g]
        # least loaded part; among near-equal loads (within one gene's cost) the first one with the fewest genes of this class
        thr = min(load) + 0.5 * cg
        r, fewest = -1, None
        for k in range(n):
            if load[k] <= thr and (fewest is None or ncls[k][c] < fewest):
                r, fewest = k, ncls[k][c]
        parts[r].append(g)
        load[r] += cg
        ncls[r][c] += 1
    return [sorted(q) for q in parts]


def measured_gene_cost(trace, lengths, p=10, class_lengths=None, downsample_rate=1):
    """
    Relative cost of a gene in the outer iteration its counters come from: the same model as predicted_gene_cost, with the counters
    the kernels return in place of the guess "one call over L columns" -- trace[:, 1] = number of nmf() calls, trace[:, 2] = active
    columns summed over them (a high-depth gene leaves after 1 call, a noisy one after 17: the length does not say which).  The cost
    of the first outer iteration predicts every later one to a correlation of 0.99999 (round 3, profiles/round3/host_overhead_2500.txt).
    """
    L = np.asarray(lengths, dtype=np.float64)
    calls, cols = np.asarray(trace[:, 1], dtype=np.float64), np.asarray(trace[:, 2], dtype=np.float64)
    if downsample_rate > 1:
        return cols + 64.0 * np.maximum(calls, 1.0)
    if class_lengths is None:
        class_lengths = P10_CLASS_LENGTHS if int(p) == 10 else (0, 0)
    if not class_lengths[0]:
        return cols / 256.0 + 4.0 * np.maximum(calls, 0.25)
    cls = gene_classes(L, class_lengths[0], class_lengths[1])
    lanes = np.where(cls == 0, 256.0, np.where(cls == 1, 128.0, 64.0))
    per_col = np.where(cls == 0, 1.25, 1.0)
    return (cols / lanes * per_col + 4.0 * np.maximum(calls, 0.25)) * (lanes / 64.0)      # a gene that leaves before its first call still costs its scan


def rebalance_moves(owner, cost, n, tol=1.002, max_moves=None):
    """
    The FEW moves that level measured loads: starting from the current owners, repeatedly take from the most loaded part the gene
    whose cost best fills the gap to the mean and give it to the least loaded part, until no part is above tol x mean (or no move
    helps).  Deterministic (every rank computes the same list from the same all-reduced costs).  Returns [(gene, src, dst), ...].
    """
    owner = np.array(owner, dtype=np.int64)
    cost = np.asarray(cost, dtype=np.float64)
    n = int(n)
    load = np.bincount(owner, weights=cost, minlength=n).astype(np.float64)
    mean = load.mean()
    if max_moves is None:
        max_moves = max(8, len(cost) // 4)
    # per part: its genes sorted by cost (ascending) for a bisect-like pick
    members = [sorted(np.flatnonzero(owner == r).tolist(), key=lambda g: (cost[g], g)) for r in range(n)]
    moves = []
    while len(moves) < max_moves:
        hi, lo = int(np.argmax(load)), int(np.argmin(load))
        if load[hi] <= mean * tol or hi == lo or not members[hi]:
            break
        want = min(load[hi] - mean, mean - load[lo])
        if want <= 0:
            break
        costs_hi = [cost[g] for g in members[hi]]
        k = int(np.searchsorted(costs_hi, want, side='right')) - 1      # the most expensive gene not above the gap
        if k < 0:
            k = 0
            if costs_hi[0] >= load[hi] - load[lo]:                     # even the cheapest gene would overshoot: done
                break
        g = members[hi].pop(k)
        members[lo].append(g)
        members[lo].sort(key=lambda q: (cost[q], q))
        load[hi] -= cost[g]; load[lo] += cost[g]
        owner[g] = lo
        moves.append((int(g), hi, lo))
    return moves


def partition_by_measured_cost(cost, n):
    """
    LPT on MEASURED per-gene costs (the cycle counters of an outer iteration, trace column 7 ... see ShardedNMFOA.redeal): most
    expensive gene first onto the least loaded part.  Returns n sorted lists of gene positions.
    """
    cost = np.asarray(cost, dtype=np.float64)
    order = np.lexsort((np.arange(len(cost)), -cost))
    parts = [[] for _ in range(int(n))]
    load = np.zeros(int(n))
    for g in order:
        r = int(np.argmin(load))
        parts[r].append(int(g))
        load[r] += cost[g]
    return [sorted(q) for q in parts]


def chunk_bounds(length, n):
    """Start offsets (plus the end) of split_into_chunks(range(length), n)."""
    if length <= 0:
        return [0]
    size = int(np.ceil(length / n))
    return list(range(0, length, size)) + [length]
