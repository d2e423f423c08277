"""
Result writer shared by GeneNMFOA.save_results and nmf_mpi.save_results -- the file formats downstream
DegNorm tools read (reference: degnorm/nmf.py:603-711, degnorm/nmf_mpi.py:448-552; consumers
data_access.py:79-104, report.py:97-113).
"""
import os
import pickle as pkl
import warnings

import numpy as np


def write_results(genes, estimates, rho, x_adj, ran_baseline_selection, gene_manifest_df, output_dir='.',
                  sample_ids=None, p=None, degnorm_iter=None):
    from pandas import DataFrame, concat

    if not os.path.isdir(output_dir):
        raise IOError('Directory {0} not found.'.format(output_dir))
    if not all(col in gene_manifest_df.columns.tolist() for col in ['chr', 'gene']):
        raise ValueError('gene_manifest_df must have columns `chr` and `gene`.')

    p = rho.shape[1] if p is None else p
    degnorm_iter = ran_baseline_selection.shape[1] if degnorm_iter is None else degnorm_iter
    if sample_ids:
        if len(sample_ids) != p:
            raise ValueError('Number of supplied sample IDs does not match number'
                             'of samples used to fit GeneNMFOA object.')
        sample_ids = list(sample_ids)
    else:
        sample_ids = ['sample_{0}'.format(i + 1) for i in range(p)]

    if isinstance(estimates, dict):
        estimates = [estimates[g] for g in genes]

    known = np.intersect1d(gene_manifest_df.gene.unique(), genes)
    if len(known) < len(genes):
        warnings.warn('Gene manifest data does not encompass set of genes sent through DegNorm.')
    if len(known) == 0:
        raise ValueError('No genes used in DegNorm were found in gene manifest dataframe!')

    manifest = gene_manifest_df[gene_manifest_df.gene.isin(known)]
    chroms = manifest.chr.unique().tolist()
    first_chr = manifest.drop_duplicates('gene').set_index('gene').chr
    position = {g: k for k, g in enumerate(genes)}

    # {chromosome: {gene: estimate}}; genes inside a chromosome follow the sorted order of `known`,
    # as in the reference (it walks np.intersect1d's output, nmf.py:654-660).
    per_chrom = {c: dict() for c in chroms}
    for g in known:
        per_chrom[first_chr[g]][g] = estimates[position[g]]

    for c in chroms:
        cdir = os.path.join(output_dir, str(c))
        if not os.path.isdir(cdir):
            os.makedirs(cdir)
        with open(os.path.join(cdir, 'estimated_coverage_matrices_{0}.pkl'.format(c)), 'wb') as f:
            pkl.dump(per_chrom[c], f)

    missing = [g for g in genes if g not in first_chr.index]
    if missing:
        raise KeyError('{0} gene(s) sent through DegNorm are missing from the gene manifest, e.g. {1}'
                       .format(len(missing), missing[0]))
    index_df = DataFrame({'chr': [first_chr[g] for g in genes], 'gene': list(genes)})

    def dump(values, columns, name):
        df = concat([index_df, DataFrame(values, columns=columns)], axis=1)
        df[['chr', 'gene'] + columns].to_csv(os.path.join(output_dir, name), index=False)

    dump(rho, sample_ids, 'degradation_index_scores.csv')
    dump(x_adj, sample_ids, 'adjusted_read_counts.csv')
    dump(ran_baseline_selection, ['iter_{0}'.format(i) for i in range(degnorm_iter)], 'ran_baseline_selection.csv')
