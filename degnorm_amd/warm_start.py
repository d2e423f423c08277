"""
Warm-start entry of the hot path: read a previous DegNorm output directory, apply the CLI's gene
filter and hand the result to GeneNMFOA -- the steps immediately before the NMF-OA core in the reference
(`degnorm/warm_start.py:10-106`, `degnorm/__main__.py:219-247`, `degnorm/__main_mpi__.py:364-394`).

On-disk layout read here (written by the reference at `reads_coverage_merge.py:446-452`, `__main__.py:199-209`):
    <dir>/gene_exon_metadata.csv          columns chr, gene, gene_start, gene_end, ...
    <dir>/read_counts.csv                 columns chr, gene, <sample ids...>
    <dir>/<chr>/coverage_matrices_<chr>.pkl   pickled dict {gene: (p x L) float64 ndarray}

    python -m degnorm_amd.warm_start --warm-start-dir DIR -o OUT [--iter 5 --nmf-iter 100 -d 1 -s --minimax-coverage 0]
"""
import argparse
import gc
import logging
import os
import pickle as pkl
import shutil
from collections import OrderedDict

import numpy as np


def load_from_previous(degnorm_dir, new_dir=None):
    """
    Same contract as the reference's `load_from_previous` (warm_start.py:10-106): returns a dict with
    `gene_cov_dict` (OrderedDict gene -> p x L matrix, genes in per-chromosome pickle order),
    `read_count_df` and `genes_df` (rows in that same gene order) and `sample_ids`.
    When `new_dir` is given the three inputs are copied there like the reference does.
    """
    from pandas import read_csv

    if new_dir is not None and not os.path.isdir(new_dir):
        raise IOError('new DegNorm output directory {0} not found.'.format(new_dir))
    exon_file = os.path.join(degnorm_dir, 'gene_exon_metadata.csv')
    read_count_file = os.path.join(degnorm_dir, 'read_counts.csv')
    exon_df = read_csv(exon_file, low_memory=False)          # FileNotFoundError propagates, as in the reference
    read_count_df = read_csv(read_count_file, low_memory=False)
    if new_dir is not None:
        shutil.copy(exon_file, os.path.join(new_dir, 'gene_exon_metadata.csv'))
        shutil.copy(read_count_file, os.path.join(new_dir, 'read_counts.csv'))

    genes_df = exon_df[['chr', 'gene', 'gene_start', 'gene_end']].drop_duplicates().reset_index(drop=True)
    keep = np.intersect1d(genes_df.gene, read_count_df.gene)             # genes known to both tables
    keep_set = set(keep.tolist())
    genes_df = genes_df[genes_df.gene.isin(keep)]
    read_count_df = read_count_df[read_count_df.gene.isin(keep)]
    sample_ids = read_count_df.columns.tolist()[2:]

    gene_cov_dict = OrderedDict()
    for chrom in genes_df.chr.unique().tolist():
        cov_file = os.path.join(degnorm_dir, str(chrom), 'coverage_matrices_{0}.pkl'.format(chrom))
        if new_dir is not None:
            os.makedirs(os.path.join(new_dir, str(chrom)))
            shutil.copy(cov_file, os.path.join(new_dir, str(chrom), 'coverage_matrices_{0}.pkl'.format(chrom)))
        with open(cov_file, 'rb') as f:
            cov_dat = pkl.load(f)
        for gene in cov_dat:
            if gene in keep_set:
                gene_cov_dict[gene] = cov_dat[gene]
        del cov_dat
    gc.collect()

    genes = list(gene_cov_dict.keys())
    genes_df = genes_df.set_index('gene').loc[genes].reset_index(drop=False)
    read_count_df = read_count_df.set_index('gene').loc[genes].reset_index(drop=False)
    return {'gene_cov_dict': gene_cov_dict, 'read_count_df': read_count_df, 'genes_df': genes_df,
            'sample_ids': sample_ids}


def select_genes(gene_cov_dict, read_count_df, genes_df, minimax_coverage=0, downsample_rate=1, mpi_limits=False):
    """
    The CLI's gene filter (`__main__.py:219-247`): drop a gene when its maximum coverage is below
    `minimax_coverage` or its length is <= `downsample_rate`; with `mpi_limits` also when it is longer than 9e6
    bases or its maximum coverage exceeds 2^31 - 1 (`__main_mpi__.py:374-376`).  Mutates `gene_cov_dict` like the
    reference and returns (gene_cov_dict, read_count_df, genes_df) with matching rows.
    """
    drop = []
    for i in range(genes_df.shape[0]):
        gene = genes_df.gene.iloc[i]
        cov = gene_cov_dict[gene]
        bad = (cov.max() < minimax_coverage) or (cov.shape[1] <= downsample_rate)
        if mpi_limits:
            bad = bad or (cov.shape[1] > 9e6) or (cov.max() > 2147483647)
        if bad:
            drop.append(i)
            del gene_cov_dict[gene]
    if drop:
        read_count_df = read_count_df.drop(drop, axis=0).reset_index(drop=True)
        genes_df = genes_df.drop(drop, axis=0).reset_index(drop=True)
    if (read_count_df.shape[0] == 0) or genes_df.empty or (len(gene_cov_dict) == 0):
        raise ValueError('No genes available to run through DegNorm!\n'
                         'Check that your requested genes are in genome annotation file.')
    if len(gene_cov_dict.keys()) != read_count_df.shape[0]:
        raise ValueError('Number of coverage matrices not equal to number of genes in read count DataFrame!')
    return gene_cov_dict, read_count_df, genes_df


def run_from_warm_start(warm_start_dir, output_dir, degnorm_iter=5, nmf_iter=100, downsample_rate=1,
                        skip_baseline_selection=False, minimax_coverage=0, device=None, model=None):
    """Warm-start directory -> filter -> GeneNMFOA.run -> save_results, as the `degnorm` CLI chains them."""
    from .nmf import GeneNMFOA
    dat = load_from_previous(warm_start_dir, output_dir)
    cov, reads_df, genes_df = select_genes(dat['gene_cov_dict'], dat['read_count_df'], dat['genes_df'],
                                           minimax_coverage=minimax_coverage, downsample_rate=downsample_rate)
    sample_ids = dat['sample_ids']
    logging.info('DegNorm will run on {0} genes, downsampling rate = 1 / {1}, {2} baseline selection.'
                 .format(len(cov), downsample_rate, 'without' if skip_baseline_selection else 'with'))
    if model is None:
        model = GeneNMFOA(degnorm_iter=degnorm_iter, nmf_iter=nmf_iter, downsample_rate=downsample_rate,
                          skip_baseline_selection=skip_baseline_selection, device=device)
    estimates = model.run(cov, reads_dat=reads_df[sample_ids].values.astype(np.float64))      # __main__.py:269-270
    model.save_results(estimates, gene_manifest_df=genes_df, output_dir=output_dir, sample_ids=sample_ids)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(description='NMF-OA core of DegNorm from a warm-start directory (MI355X)')
    ap.add_argument('--warm-start-dir', required=True)
    ap.add_argument('-o', '--output-dir', required=True)
    ap.add_argument('--iter', type=int, default=5)
    ap.add_argument('--nmf-iter', type=int, default=100)
    ap.add_argument('-d', '--downsample-rate', type=int, default=1)
    ap.add_argument('-s', '--skip-baseline-selection', action='store_true')
    ap.add_argument('--minimax-coverage', type=int, default=0)
    ap.add_argument('--device', type=int, default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s ---- %(message)s')
    os.makedirs(args.output_dir, exist_ok=True)
    run_from_warm_start(args.warm_start_dir, args.output_dir, degnorm_iter=args.iter, nmf_iter=args.nmf_iter,
                        downsample_rate=args.downsample_rate, skip_baseline_selection=args.skip_baseline_selection,
                        minimax_coverage=args.minimax_coverage, device=args.device)


if __name__ == '__main__':
    main()
