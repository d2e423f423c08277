"""
Seeded synthetic coverage generator.

Produces per-gene coverage matrices (p x L_g, non-negative integer counts) and the matching
read-count matrix, in the shape the hot path consumes (reference input contract:
degnorm/reads_coverage_merge.py:353, degnorm/__main__.py:269-270).

Every gene g draws from its own generator ``default_rng([seed, g])`` so that any subset or any
rank's shard of a configuration can be generated independently and identically (the GPU box,
the CPU container and every rank of a multi-GPU run see bit-identical inputs).

Gene classes (fractions committed here; they exercise every exit of baseline selection,
degnorm/nmf.py:232,241,257,265,273,327,342,349):
    0 global ramp 30 %, 1 partial ramp 20 %, 2 none 10 %, 3 high-depth none 15 %,
    4 high-depth partial 10 %, 5 low coverage 5 %, 6 short window 4 %, 7 spike 3 %, 8 dropout 3 %.
"""
import numpy as np
from collections import OrderedDict

CLASS_NAMES = ('global_ramp', 'partial_ramp', 'none', 'hd_none', 'hd_partial',
               'low_cov', 'short_window', 'spike', 'dropout')
CLASS_FRAC = np.array([0.30, 0.20, 0.10, 0.15, 0.10, 0.05, 0.04, 0.03, 0.03])
_CLASS_CDF = np.cumsum(CLASS_FRAC)

# configuration presets from BASELINE.json `configs` (seed per config: BASELINE.md section 4).
CONFIGS = {
    'c1': dict(seed=1, n_genes=100, p=4, l_min=1000, l_max=1000),
    'c2': dict(seed=2, n_genes=20000, p=10, l_min=200, l_max=5000),
    'c4': dict(seed=4, n_genes=50000, p=50, l_min=501, l_max=5000),
    'c5': dict(seed=5, n_genes=2000, p=6, l_min=200, l_max=5000),
}


def gene_length(seed, g, l_min=200, l_max=5000):
    """Length of gene g (first draw of the gene's stream) without generating its coverage."""
    rng = np.random.default_rng([int(seed), int(g)])
    return int(rng.integers(l_min, l_max + 1))


def synth_gene(seed, g, p, l_min=200, l_max=5000, dtype=np.float64):
    """
    Coverage matrix of gene g: (p x L) array of Poisson counts.

    :return: (coverage (p x L) ndarray, class id int)
    """
    rng = np.random.default_rng([int(seed), int(g)])
    L = int(rng.integers(l_min, l_max + 1))
    cls = int(np.searchsorted(_CLASS_CDF, rng.random(), side='right'))
    cls = min(cls, len(CLASS_FRAC) - 1)

    j = np.arange(L, dtype=np.float64)
    a = rng.uniform(20., 100.)
    f = rng.uniform(1., 3.)
    env = 20. + a * np.abs(np.sin(np.pi * f * j / L))
    abund = rng.lognormal(0., 0.5, size=p)
    depth = rng.uniform(30., 100.) if cls in (3, 4) else 1.

    deg = np.ones((p, L))
    if cls == 0:
        for i in range(p):
            if rng.random() < 0.5:
                deg[i] = np.linspace(rng.uniform(0.2, 0.9), 1., L)
    elif cls in (1, 4):
        w = max(2, int(L * rng.uniform(0.3, 0.5)))
        for i in range(p):
            if rng.random() < 0.5:
                deg[i, :w] = np.linspace(rng.uniform(0.2, 0.9), 1., w)

    mean = depth * abund[:, None] * env[None, :] * deg

    if cls == 5:
        mean *= 0.02
    elif cls in (6, 7):
        width = int(rng.integers(60, 191)) if cls == 6 else int(rng.integers(10, 46))
        width = min(width, L)
        start = int(rng.integers(0, L - width + 1))
        mask = np.full(L, 0.01 if cls == 6 else 0.002)
        mask[start:start + width] = 1.
        mean *= mask[None, :]
    elif cls == 8:
        mean[int(rng.integers(0, p))] = 0.

    cov = rng.poisson(mean).astype(dtype)

    # no all-zero genes (reference raises ArpackError on them: SURVEY H8).
    if not cov.any():
        cov[0, 0] = 1.

    return cov, cls


def pileup_gene(seed, g, p, l_min=300, l_max=3000, dtype=np.float64):
    """
    Read pile-up coverage of gene g: what DegNorm's real input looks like (reads.py:714,773 -- every base of a read adds 1 to the
    coverage of the positions it spans), as opposed to synth_gene's envelope x Poisson draw: reads of 75-150 bases stacked at low to
    medium depth, so the coverage is a piecewise-constant small integer; per sample a 3' bias (start density rising towards the 3'
    end by a random exponent: degradation), optionally uneven exon depth and a stretch without reads.  Genes of this kind have
    exact ties everywhere (10 x == max on integer counts, equal bin means): the kind the round-3 fuzz run called `steps`.

    :return: (coverage (p x L) ndarray of whole numbers, kind id: 0 plain, 1 3' decay, 2 decay + uneven exons, 3 decay + a gap)
    """
    rng = np.random.default_rng([int(seed), int(g), 77])
    L = int(rng.integers(l_min, l_max + 1))
    kind = int(rng.integers(0, 4))
    depth = float(rng.choice([2., 5., 12., 30.]))                 # mean coverage of the best-covered sample
    abund = rng.lognormal(0., 0.5, size=p)
    abund /= abund.max()
    pos = (np.arange(L) + 0.5) / L
    exon = np.ones(L)
    if kind == 2:                                                  # uneven exon usage: piecewise-constant multipliers
        cuts = np.sort(rng.integers(1, L, size=int(rng.integers(1, 5))))
        lv = rng.uniform(0.3, 1.0, size=len(cuts) + 1)
        exon = lv[np.searchsorted(cuts, np.arange(L), side='right')]
    cov = np.zeros((p, L))
    for i in range(p):
        dens = exon.copy()
        if kind >= 1 and rng.random() < 0.7:
            dens = dens * (0.05 + pos) ** rng.uniform(0.3, 2.5)   # 3' bias of a degraded sample
        if kind == 3:
            a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(20, max(21, L // 4))))
            dens[a:b] = 0.
        if not dens.any():
            dens[:] = 1.
        cdf = np.cumsum(dens); cdf /= cdf[-1]
        n_reads = int(rng.poisson(depth * abund[i] * L / 112.))
        mid = np.searchsorted(cdf, rng.random(n_reads))            # read midpoints follow the density
        rl = rng.integers(75, 151, size=n_reads)
        lo = np.clip(mid - rl // 2, 0, L)
        hi = np.clip(lo + rl, 0, L)
        d = np.zeros(L + 1)
        np.add.at(d, lo, 1.); np.add.at(d, hi, -1.)
        cov[i] = np.cumsum(d[:L])
    if not cov.any():
        cov[0, :min(L, 100)] = 1.
    return cov.astype(dtype), kind


def pileup_dataset(seed, gene_ids, p, l_min=300, l_max=3000, dtype=np.float64):
    """pile-up genes in the shape synth_dataset returns; read counts = reads whose midpoint falls in the gene ~ rowsum / 112"""
    cov_dat = OrderedDict()
    gene_ids = list(gene_ids)
    reads = np.zeros((len(gene_ids), p))
    kinds = np.zeros(len(gene_ids), dtype=np.int32)
    for k, g in enumerate(gene_ids):
        cov, kind = pileup_gene(seed, g, p, l_min, l_max, dtype=dtype)
        cov_dat['pileup_{0:06d}'.format(g)] = cov
        reads[k] = np.round(cov.sum(axis=1, dtype=np.float64) / 112.)
        kinds[k] = kind
    return cov_dat, reads, kinds


def read_counts_from_coverage(cov):
    """Read counts of a gene: round(rowsum / 100)."""
    return np.round(cov.sum(axis=1, dtype=np.float64) / 100.)


def synth_dataset(seed, n_genes, p, l_min=200, l_max=5000, gene_ids=None, dtype=np.float64):
    """
    Build the hot path's inputs for genes ``gene_ids`` (default: all ``n_genes``) of a configuration.

    :return: (OrderedDict {gene name: (p x L) ndarray}, reads (n x p) float64, classes (n,) int)
    """
    if gene_ids is None:
        gene_ids = range(n_genes)
    gene_ids = list(gene_ids)
    cov_dat = OrderedDict()
    reads = np.zeros((len(gene_ids), p))
    classes = np.zeros(len(gene_ids), dtype=np.int32)
    for k, g in enumerate(gene_ids):
        cov, cls = synth_gene(seed, g, p, l_min, l_max, dtype=dtype)
        cov_dat['gene_{0:06d}'.format(g)] = cov
        reads[k] = read_counts_from_coverage(cov)
        classes[k] = cls
    return cov_dat, reads, classes


def synth_packed(seed, gene_ids, p, l_min=200, l_max=5000, n_threads=8):
    """
    Generate genes straight into the packed fp32 layout the device consumes
    (sample-major p x L_g per gene, genes back to back; see DESIGN.md "Data layout").

    :return: (packed float32 1-d array, lengths int64 (n,), reads float64 (n x p), classes int32 (n,))
    """
    from concurrent.futures import ThreadPoolExecutor
    gene_ids = list(gene_ids)
    n = len(gene_ids)
    lengths = np.array([gene_length(seed, g, l_min, l_max) for g in gene_ids], dtype=np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lengths * p, out=offs[1:])
    packed = np.empty(int(offs[-1]), dtype=np.float32)
    reads = np.zeros((n, p))
    classes = np.zeros(n, dtype=np.int32)

    def work(k):
        cov, cls = synth_gene(seed, gene_ids[k], p, l_min, l_max)
        packed[offs[k]:offs[k + 1]] = cov.reshape(-1)
        reads[k] = read_counts_from_coverage(cov)
        classes[k] = cls

    if n_threads > 1 and n > 64:
        with ThreadPoolExecutor(max_workers=n_threads) as ex:
            list(ex.map(work, range(n), chunksize=64))
    else:
        for k in range(n):
            work(k)
    return packed, lengths, reads, classes
