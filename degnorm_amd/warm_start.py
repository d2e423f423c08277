"""
Warm-start entry of the hot path: read a previous DegNorm output directory, apply the CLI's gene
filter and hand the result to GeneNMFOA -- the steps immediately before the NMF-OA core in the reference
(`degnorm/warm_start.py:10-106`, `degnorm/__main__.py:219-247`, `degnorm/__main_mpi__.py:364-394`).

On-disk layout read here (written by the reference at `reads_coverage_merge.py:446-452`, `__main__.py:199-209`):
    <dir>/gene_exon_metadata.csv          columns chr, gene, gene_start, gene_end, ...
    <dir>/read_counts.csv                 columns chr, gene, <sample ids...>
    <dir>/<chr>/coverage_matrices_<chr>.pkl   pickled dict {gene: (p x L) float64 ndarray}

Optional packed side-car (written by `write_sidecars`, never by the reference; ignored when stale):
    <dir>/<chr>/coverage_matrices_<chr>.f32.npy       the chromosome's matrices back to back as float32 -- the layout the
                                                      device consumes (gene g: p rows of L_g, genes in pickle order)
    <dir>/<chr>/coverage_matrices_<chr>.f32.idx.npz   gene names, lengths, p, size / mtime of the pickle it was made from
Unpickling 4 GB of float64 dominates a warm start of a human data set; the side-car is half the bytes, memory-mapped, and
hands out float32 views that the uploader copies without conversion (profiles/round2/warm_start_load.txt).

    python -m degnorm_amd.warm_start --warm-start-dir DIR -o OUT [--iter 5 --nmf-iter 100 -d 1 -s --minimax-coverage 0]
    python -m degnorm_amd.warm_start --warm-start-dir DIR --write-sidecars
    python -m torch.distributed.run --nproc-per-node N -m degnorm_amd.warm_start --mpi --warm-start-dir DIR -o OUT ...
"""
import argparse
import gc
import logging
import os
import pickle as pkl
import shutil
from collections import OrderedDict

import numpy as np


def _sidecar_paths(degnorm_dir, chrom):
    base = os.path.join(degnorm_dir, str(chrom), 'coverage_matrices_{0}'.format(chrom))
    return base + '.pkl', base + '.f32.npy', base + '.f32.idx.npz'


def write_sidecars(degnorm_dir, chroms=None):
    """
    Write the packed float32 side-car of every `<chr>/coverage_matrices_<chr>.pkl` under `degnorm_dir` (once per
    DegNorm output directory; later warm starts read it instead of the pickle).  Returns {chrom: number of values that
    float32 cannot hold exactly} -- 0 for DegNorm's integer coverage counts (reads.py:714,773).
    """
    if chroms is None:
        chroms = sorted(d for d in os.listdir(degnorm_dir)
                        if os.path.isfile(_sidecar_paths(degnorm_dir, d)[0]))
    out = dict()
    for chrom in chroms:
        pkl_file, npy_file, idx_file = _sidecar_paths(degnorm_dir, chrom)
        with open(pkl_file, 'rb') as f:
            cov_dat = pkl.load(f)
        genes = list(cov_dat.keys())
        lengths = np.array([cov_dat[g].shape[1] for g in genes], dtype=np.int64)
        p = int(cov_dat[genes[0]].shape[0]) if genes else 0
        packed = np.lib.format.open_memmap(npy_file, mode='w+', dtype=np.float32, shape=(int(p * lengths.sum()),))
        o, inexact = 0, 0
        maxcov = np.zeros(len(genes))
        for k, (g, L) in enumerate(zip(genes, lengths)):
            m = np.asarray(cov_dat[g])
            blk = packed[o:o + p * int(L)].reshape(p, int(L))
            blk[...] = m
            inexact += int(np.count_nonzero(blk != m))
            maxcov[k] = float(m.max())                     # what the CLI's gene filter asks of a gene (__main__.py:229): no need to read it again
            o += p * int(L)
        packed.flush()
        del packed
        st = os.stat(pkl_file)
        np.savez(idx_file, genes=np.array(genes, dtype=str), lengths=lengths, p=p, inexact=inexact, maxcov=maxcov,
                 pkl_size=st.st_size, pkl_mtime_ns=st.st_mtime_ns)
        out[chrom] = inexact
    return out


def _read_sidecar(pkl_file, npy_file, idx_file):
    """
    float32 views of the memory-mapped side-car, or None (with a log line) when it cannot be trusted: made from another
    pickle (size / mtime), an index that does not describe the data file (p * sum(lengths) values, one length per gene), or
    anything unreadable -- the caller then falls back to the pickle.
    """
    try:
        with np.load(idx_file) as idx:
            genes, lengths = idx['genes'].tolist(), idx['lengths'].astype(np.int64)
            p, inexact = int(idx['p']), int(idx['inexact'])
            pkl_size, pkl_mtime_ns = int(idx['pkl_size']), int(idx['pkl_mtime_ns'])
        st = os.stat(pkl_file)
        if pkl_size != st.st_size or pkl_mtime_ns != st.st_mtime_ns:
            logging.info('{0} is stale (the pickle changed): ignored.'.format(npy_file))
            return None
        packed = np.load(npy_file, mmap_mode='r')
        if (packed.dtype != np.float32 or packed.ndim != 1 or len(genes) != len(lengths) or p < 1
                or np.any(lengths < 1) or packed.size != p * int(lengths.sum())):
            logging.warning('{0} does not match its index ({1} values for p = {2}, {3} genes): ignored.'
                            .format(npy_file, packed.size, p, len(genes)))
            return None
    except Exception as e:                                     # truncated / foreign files: the pickle is the source of truth
        logging.warning('{0} could not be read ({1}): ignored.'.format(npy_file, e))
        return None
    if inexact:
        logging.warning('{0}: {1} coverage values are not exactly representable in float32.'.format(npy_file, inexact))
    out, o = OrderedDict(), 0
    for g, L in zip(genes, lengths.tolist()):
        out[g] = packed[o:o + p * L].reshape(p, L)
        o += p * L
    return out


def _sidecar_index(degnorm_dir, chrom):
    """
    The side-car's index alone (gene names, lengths, per-gene maximum, p) when the side-car can be trusted -- made from the pickle
    that is there now, describing the data file that is there now, written with the per-gene maxima -- else None.  No coverage is read.
    """
    pkl_file, npy_file, idx_file = _sidecar_paths(degnorm_dir, chrom)
    try:
        with np.load(idx_file) as idx:
            if 'maxcov' not in idx.files:
                return None
            genes, lengths, maxcov = idx['genes'].tolist(), idx['lengths'].astype(np.int64), idx['maxcov'].astype(np.float64)
            p, pkl_size, pkl_mtime_ns = int(idx['p']), int(idx['pkl_size']), int(idx['pkl_mtime_ns'])
        st = os.stat(pkl_file)
        if pkl_size != st.st_size or pkl_mtime_ns != st.st_mtime_ns:
            return None
        if os.stat(npy_file).st_size < 4 * p * int(lengths.sum()) or len(genes) != len(lengths) or len(maxcov) != len(genes) or p < 1:
            return None
    except Exception:
        return None
    offs = np.zeros(len(genes) + 1, dtype=np.int64)
    np.cumsum(p * lengths, out=offs[1:])
    return dict(genes=genes, lengths=lengths, maxcov=maxcov, p=p, offsets=offs[:-1], npy_file=npy_file)


def load_index_from_previous(degnorm_dir):
    """
    load_from_previous WITHOUT the coverage: the two tables and, per gene (in the reference's order: per-chromosome pickle order,
    warm_start.py:59-97), its chromosome, length, maximum coverage and position in the chromosome's packed side-car.  None unless
    every chromosome has an up-to-date side-car with per-gene maxima (write_sidecars).  What a rank of a sharded run needs to
    pick and memory-map ITS genes without anybody loading the whole data set.
    """
    from pandas import read_csv
    exon_df = read_csv(os.path.join(degnorm_dir, 'gene_exon_metadata.csv'), low_memory=False)
    read_count_df = read_csv(os.path.join(degnorm_dir, 'read_counts.csv'), low_memory=False)
    genes_df = exon_df[['chr', 'gene', 'gene_start', 'gene_end']].drop_duplicates().reset_index(drop=True)
    keep = set(np.intersect1d(genes_df.gene, read_count_df.gene).tolist())
    genes_df = genes_df[genes_df.gene.isin(keep)]
    read_count_df = read_count_df[read_count_df.gene.isin(keep)]
    sample_ids = read_count_df.columns.tolist()[2:]
    names, chrom_of, lengths, maxcov, offsets, files, p = [], [], [], [], [], {}, None
    for chrom in genes_df.chr.unique().tolist():
        ix = _sidecar_index(degnorm_dir, chrom)
        if ix is None or (p is not None and ix['p'] != p):
            return None
        p = ix['p']
        files[str(chrom)] = ix['npy_file']
        for k, g in enumerate(ix['genes']):
            if g in keep:
                names.append(g); chrom_of.append(str(chrom)); lengths.append(int(ix['lengths'][k]))
                maxcov.append(float(ix['maxcov'][k])); offsets.append(int(ix['offsets'][k]))
    if not names:
        return None
    genes_df = genes_df.set_index('gene').loc[names].reset_index(drop=False)
    read_count_df = read_count_df.set_index('gene').loc[names].reset_index(drop=False)
    return dict(genes=names, chrom=chrom_of, lengths=np.array(lengths, dtype=np.int64), maxcov=np.array(maxcov), offsets=np.array(offsets, dtype=np.int64),
                files=files, p=int(p), read_count_df=read_count_df, genes_df=genes_df, sample_ids=sample_ids)


def pack_genes_from_sidecars(index, positions):
    """One packed float32 buffer (the layout dn_upload_packed takes) of the genes at `positions` of a load_index_from_previous
    result, copied out of the memory-mapped side-cars: only these genes' pages are read."""
    p = index['p']
    lengths = index['lengths'][np.asarray(positions, dtype=np.int64)] if len(positions) else np.zeros(0, dtype=np.int64)
    packed = np.empty(int(p * lengths.sum()), dtype=np.float32)
    maps, o = {}, 0
    for k in positions:
        c = index['chrom'][k]
        if c not in maps:
            maps[c] = np.load(index['files'][c], mmap_mode='r')
        n = p * int(index['lengths'][k])
        packed[o:o + n] = maps[c][index['offsets'][k]:index['offsets'][k] + n]
        o += n
    return packed, lengths


def _load_chrom(degnorm_dir, chrom, use_sidecar):
    """{gene: p x L matrix} of one chromosome: float32 views into the memory-mapped side-car when it is present and was
    made from the pickle that is there now, else the unpickled float64 dict."""
    pkl_file, npy_file, idx_file = _sidecar_paths(degnorm_dir, chrom)
    if use_sidecar and os.path.isfile(npy_file) and os.path.isfile(idx_file):
        out = _read_sidecar(pkl_file, npy_file, idx_file)
        if out is not None:
            return out
    with open(pkl_file, 'rb') as f:
        return pkl.load(f)


def load_from_previous(degnorm_dir, new_dir=None, use_sidecar=True):
    """
    Same contract as the reference's `load_from_previous` (warm_start.py:10-106): returns a dict with
    `gene_cov_dict` (OrderedDict gene -> p x L matrix, genes in per-chromosome pickle order),
    `read_count_df` and `genes_df` (rows in that same gene order) and `sample_ids`.
    When `new_dir` is given the three inputs are copied there like the reference does.
    `use_sidecar` (extra): read `coverage_matrices_<chr>.f32.npy` instead of the pickle when it is up to date; the
    matrices are then float32 (exact for coverage counts) views of a memory map.
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
        cov_dat = _load_chrom(degnorm_dir, chrom, use_sidecar)
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


def _copy_inputs(degnorm_dir, new_dir, chroms):
    """what load_from_previous copies into the new output directory (warm_start.py:30-57)"""
    shutil.copy(os.path.join(degnorm_dir, 'gene_exon_metadata.csv'), os.path.join(new_dir, 'gene_exon_metadata.csv'))
    shutil.copy(os.path.join(degnorm_dir, 'read_counts.csv'), os.path.join(new_dir, 'read_counts.csv'))
    for chrom in chroms:
        os.makedirs(os.path.join(new_dir, str(chrom)), exist_ok=True)
        shutil.copy(os.path.join(degnorm_dir, str(chrom), 'coverage_matrices_{0}.pkl'.format(chrom)),
                    os.path.join(new_dir, str(chrom), 'coverage_matrices_{0}.pkl'.format(chrom)))


def run_from_warm_start_mpi(comm, warm_start_dir, output_dir, degnorm_iter=5, nmf_iter=100, downsample_rate=1,
                            skip_baseline_selection=False, minimax_coverage=0, device=None, partition='balanced',
                            sharded_load=True):
    """
    The `degnorm_mpi --warm-start-dir` chain (`__main_mpi__.py:357-456`) on one process per GPU.
    With up-to-date packed side-cars (write_sidecars) NO rank loads the whole data set: every rank reads the two tables and the
    side-car indices (names, lengths, per-gene maxima), applies the MPI CLI's gene filter to them (minimax coverage, take-every
    size, its 9-megabase / 2^31 limits, :374-376), derives the same partition, and memory-maps ITS genes (the reference has rank 0
    load everything and every worker unpickle the WHOLE dictionary from a temporary file, :400-415).  Without side-cars rank 0
    loads the pickles and run_gene_nmfoa_mpi ships every rank its packed share once.  Rank 0 writes the result files (:450-456).
    A failure before the run (missing files, no genes left) is raised on every rank.  Returns the result dict on rank 0, else None.
    """
    from .nmf_mpi import run_gene_nmfoa_mpi, save_results, _bcast, _allreduce, _partition, _run_shard_and_gather
    index, err = None, None
    if sharded_load:
        try:
            index = load_index_from_previous(warm_start_dir)
        except (IOError, OSError, ValueError, KeyError) as e:
            index, err = None, '{0}: {1}'.format(type(e).__name__, e)
    n_ok = int(round(_allreduce(comm, [1.0 if index is not None else 0.0])[0]))
    if n_ok == comm.size:
        try:
            L, mx = index['lengths'], index['maxcov']
            bad = (mx < minimax_coverage) | (L <= downsample_rate) | (L > 9e6) | (mx > 2147483647)      # __main__.py:229, __main_mpi__.py:374-376
            keep = np.flatnonzero(~bad)
            if len(keep) == 0:
                raise ValueError('No genes available to run through DegNorm!\n'
                                 'Check that your requested genes are in genome annotation file.')
            sample_ids = index['sample_ids']
            genes = [index['genes'][k] for k in keep]
            genes_df = index['genes_df'].iloc[keep].reset_index(drop=True)
            x = index['read_count_df'].iloc[keep][sample_ids].values.astype(np.float64)          # __main_mpi__.py:430
            p, li_vec = index['p'], L[keep]
            if abs(int(downsample_rate)) > 1 and not np.min(li_vec) >= abs(int(downsample_rate)):
                raise ValueError('downsample_rate is too large; take-every size > at least one gene.')
            parts = _partition(li_vec, comm.size, p, downsample_rate, partition, device)           # the same on every rank
            if comm.rank == 0:
                _copy_inputs(warm_start_dir, output_dir, list(dict.fromkeys(index['chrom'])))
                logging.info('DegNorm will run on {0} genes, downsampling rate = 1 / {1}, {2} baseline selection; every rank '
                             'maps its own genes from the packed side-cars.'.format(len(genes), downsample_rate,
                                                                                    'without' if skip_baseline_selection else 'with'))
            mine = parts[comm.rank]
            packed, lengths = pack_genes_from_sidecars(index, [int(keep[k]) for k in mine])
        except (IOError, OSError, ValueError, KeyError) as e:
            err = '{0}: {1}'.format(type(e).__name__, e)
        n_bad = int(round(_allreduce(comm, [1.0 if err is not None else 0.0])[0]))
        if n_bad:
            raise ValueError('warm start failed -- ' + (err or 'on another rank'))
        eng_kw = dict(device=device, degnorm_iter=degnorm_iter, downsample_rate=downsample_rate, nmf_iter=nmf_iter,
                      skip_baseline_selection=skip_baseline_selection)
        res = _run_shard_and_gather(comm, eng_kw, [genes[k] for k in mine], packed, lengths, x[mine], np.asarray(mine, dtype=np.int64),
                                    len(genes), p, degnorm_iter, genes, li_vec, parts)
    else:
        err, cov, reads, genes_df, sample_ids = None, None, None, None, None
        if comm.rank == 0:
            try:
                dat = load_from_previous(warm_start_dir, output_dir)
                cov, reads_df, genes_df = select_genes(dat['gene_cov_dict'], dat['read_count_df'], dat['genes_df'],
                                                       minimax_coverage=minimax_coverage, downsample_rate=downsample_rate,
                                                       mpi_limits=True)
                sample_ids = dat['sample_ids']
                reads = reads_df[sample_ids].values.astype(np.float64)            # __main_mpi__.py:430
                logging.info('DegNorm will run on {0} genes, downsampling rate = 1 / {1}, {2} baseline selection.'
                             .format(len(cov), downsample_rate, 'without' if skip_baseline_selection else 'with'))
            except (IOError, OSError, ValueError, KeyError) as e:
                err = '{0}: {1}'.format(type(e).__name__, e)
        err = _bcast(comm, err)
        if err is not None:
            raise ValueError('warm start failed on rank 0 -- ' + err)
        res = run_gene_nmfoa_mpi(comm, cov, reads, degnorm_iter=degnorm_iter, nmf_iter=nmf_iter, downsample_rate=downsample_rate,
                                 skip_baseline_selection=skip_baseline_selection, device=device, partition=partition)
    if comm.rank == 0:
        save_results(genes_df, estimates=res['estimates'], rho=res['rho'], x_adj=res['x_adj'],
                     ran_baseline_selection=res['ran_baseline_selection'], sample_ids=sample_ids, output_dir=output_dir)
    comm.Barrier()
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description='NMF-OA core of DegNorm from a warm-start directory (MI355X)')
    ap.add_argument('--warm-start-dir', required=True)
    ap.add_argument('-o', '--output-dir', default=None)
    ap.add_argument('--write-sidecars', action='store_true', help='only write the packed float32 side-cars of the directory')
    ap.add_argument('--mpi', action='store_true', help='one process per GPU under torch.distributed.run (degnorm_mpi counterpart)')
    ap.add_argument('--iter', type=int, default=5)
    ap.add_argument('--nmf-iter', type=int, default=100)
    ap.add_argument('-d', '--downsample-rate', type=int, default=1)
    ap.add_argument('-s', '--skip-baseline-selection', action='store_true')
    ap.add_argument('--minimax-coverage', type=int, default=0)
    ap.add_argument('--device', type=int, default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s ---- %(message)s')
    if args.write_sidecars:
        print(write_sidecars(args.warm_start_dir))
        return
    if args.output_dir is None:
        ap.error('-o/--output-dir is required')
    if args.mpi:
        import torch
        import torch.distributed as dist
        from .nmf_mpi import TorchComm
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl' if use_gpu else 'gloo',
                                **({'device_id': torch.device('cuda', local_rank)} if use_gpu else {}))
        try:
            comm = TorchComm()
            if comm.rank == 0:
                os.makedirs(args.output_dir, exist_ok=True)
            comm.Barrier()
            run_from_warm_start_mpi(comm, args.warm_start_dir, args.output_dir, degnorm_iter=args.iter, nmf_iter=args.nmf_iter,
                                    downsample_rate=args.downsample_rate, skip_baseline_selection=args.skip_baseline_selection,
                                    minimax_coverage=args.minimax_coverage, device=args.device if args.device is not None else local_rank)
        finally:
            dist.destroy_process_group()
        return
    os.makedirs(args.output_dir, exist_ok=True)
    run_from_warm_start(args.warm_start_dir, args.output_dir, degnorm_iter=args.iter, nmf_iter=args.nmf_iter,
                        downsample_rate=args.downsample_rate, skip_baseline_selection=args.skip_baseline_selection,
                        minimax_coverage=args.minimax_coverage, device=args.device)


if __name__ == '__main__':
    main()
