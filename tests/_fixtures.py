"""
Test fixtures written to disk in the reference's on-disk layouts (test tooling, not product code): a synthetic previous-run
directory for the warm-start reader and a directory of per-sample chromosome coverage vectors for the coverage merge.
Used by tests/ and by tests/golden/make_golden.py (which feeds the same directories to the real reference).
"""
from collections import OrderedDict

import numpy as np

from degnorm_amd.synth import synth_gene, read_counts_from_coverage


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
