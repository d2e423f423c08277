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


def write_warm_start_dir(path, seed=5, n_genes=60, p=6, l_min=200, l_max=1500, chroms=('chr1', 'chr2', 'chrX')):
    """
    Write a synthetic previous-DegNorm-run directory in the reference's warm-start layout
    (degnorm/warm_start.py:30-76; files written by reads_coverage_merge.py:446-452 and __main__.py:199-209):
    gene_exon_metadata.csv, read_counts.csv, <chr>/coverage_matrices_<chr>.pkl.  Deliberately awkward:
    two exon rows per gene (duplicates), CSV row order != pickle order, one gene only in the exon table, one only
    in the read counts, one only in a pickle, and a near-empty gene (max coverage 2) for the minimax filter.
    Returns the sample ids.
    """
    import os
    import pickle
    import pandas as pd
    sample_ids = ['S{0}'.format(i + 1) for i in range(p)]
    per_chrom = {c: OrderedDict() for c in chroms}
    rows_exon, rows_reads = [], []
    for g in range(n_genes):
        cov, _ = synth_gene(seed, g, p, l_min, l_max)
        name = 'GENE{0:04d}'.format(g)
        chrom = chroms[g % len(chroms)]
        if g == 7:
            cov = np.minimum(cov, 2.0)                       # fails --minimax-coverage 5
        per_chrom[chrom][name] = cov
        L = cov.shape[1]
        start = 1000 + 10000 * g
        rows_exon.append(dict(chr=chrom, gene=name, gene_start=start, gene_end=start + L + 50, start=start, end=start + L // 2))
        rows_exon.append(dict(chr=chrom, gene=name, gene_start=start, gene_end=start + L + 50, start=start + L // 2 + 50, end=start + L + 50))
        rows_reads.append(dict(chr=chrom, gene=name, **{s: float(v) for s, v in zip(sample_ids, read_counts_from_coverage(cov))}))
    # pickle order: reversed within each chromosome
    for c in chroms:
        per_chrom[c] = OrderedDict(reversed(list(per_chrom[c].items())))
    extra, _ = synth_gene(seed, n_genes + 1, p, l_min, l_max)
    per_chrom[chroms[0]]['ORPHAN_PKL'] = extra                                              # only in a pickle
    rows_exon.append(dict(chr=chroms[1], gene='ORPHAN_EXON', gene_start=5, gene_end=900, start=5, end=900))   # only in exon table
    rows_reads.append(dict(chr=chroms[2], gene='ORPHAN_READS', **{s: 3.0 for s in sample_ids}))               # only in read counts
    rng = np.random.default_rng([int(seed), 999])
    exon_df = pd.DataFrame(rows_exon).sample(frac=1.0, random_state=int(rng.integers(1 << 30))).reset_index(drop=True)
    reads_df = pd.DataFrame(rows_reads).sample(frac=1.0, random_state=int(rng.integers(1 << 30))).reset_index(drop=True)
    os.makedirs(path, exist_ok=True)
    exon_df.to_csv(os.path.join(path, 'gene_exon_metadata.csv'), index=False)
    reads_df[['chr', 'gene'] + sample_ids].to_csv(os.path.join(path, 'read_counts.csv'), index=False)
    for c in chroms:
        os.makedirs(os.path.join(path, c), exist_ok=True)
        with open(os.path.join(path, c, 'coverage_matrices_{0}.pkl'.format(c)), 'wb') as f:
            pickle.dump(dict(per_chrom[c]), f)
    return sample_ids


def write_chrom_coverage_dir(path, seed=8, n_samples=4, chrom='chr7', chrom_len=60000, n_genes=25, missing=(2,)):
    """
    Synthetic input of the coverage-merge step in the reference's layout (reads.py:785-786,
    reads_coverage_merge.py:185-190): <path>/<sample>/chrom_coverage_<sample>_<chr>.npz holding a 1 x chrom_len CSR row
    of integer coverage, and an exon table (chr, gene, gene_start, gene_end, start, end; 1-based inclusive) with
    multi-exon genes, overlapping exons and ties in gene_end.  Samples listed in `missing` get no file (imputed zeros).
    Returns (sample_ids, exon DataFrame).
    """
    import os
    import pandas as pd
    from scipy import sparse
    rng = np.random.default_rng([int(seed), 4242])
    sample_ids = ['smp{0}'.format(i) for i in range(n_samples)]
    rows = []
    pos = 500
    for g in range(n_genes):
        n_ex = int(rng.integers(1, 5))
        gstart = pos
        exons = []
        for _ in range(n_ex):
            ln = int(rng.integers(40, 600))
            exons.append((pos, pos + ln - 1))
            pos += ln + int(rng.integers(-60, 400))           # negative gap: overlapping exons
            pos = max(pos, gstart + 1)
        gend = max(e for _, e in exons)
        for (a, b) in exons:
            rows.append(dict(chr=chrom, gene='G{0:03d}'.format(g), gene_start=gstart, gene_end=gend, start=a, end=b))
        pos = gend + int(rng.integers(50, 900))
        if pos > chrom_len - 3000:
            break
    exon_df = pd.DataFrame(rows).sample(frac=1.0, random_state=int(rng.integers(1 << 30))).reset_index(drop=True)
    os.makedirs(path, exist_ok=True)
    for i, s in enumerate(sample_ids):
        if i in missing:
            continue
        dense = rng.poisson(6.0 * (1 + i), size=chrom_len) * (rng.random(chrom_len) < 0.6)
        os.makedirs(os.path.join(path, s), exist_ok=True)
        sparse.save_npz(os.path.join(path, s, 'chrom_coverage_{0}_{1}.npz'.format(s, chrom)),
                        sparse.csr_matrix(dense.astype(int).reshape(1, -1)))
    return sample_ids, exon_df
