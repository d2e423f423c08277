"""
Coverage-matrix assembly on the device -- the producer of the NMF-OA core's input.

Reference: `degnorm/reads_coverage_merge.py:167-372` `merge_chrom_coverage` joins the per-sample chromosome
coverage vectors (`<data_dir>/<sample>/chrom_coverage_<sample>_<chr>.npz`, a 1 x N scipy CSR row written at
`reads.py:785-786`) into `{gene: (p x L) float64}` by densifying ~500 MB blocks and slicing exon ranges in Python.
Here the CSR rows go to the GPU, are expanded there, and every gene's exon union is gathered straight into the
packed fp32 layout the NMF-OA kernels read (`dn_assemble_coverage`, csrc/dn_assemble.hip).

    merge_chrom_coverage(data_dir, sample_ids, chrom_exon_df)   same signature and return value as the reference
    assemble_chrom_packed(...)                                  (genes, packed fp32, lengths) for Device.upload_packed
"""
import ctypes
import logging
import os

import numpy as np

from . import _lib


def gene_intervals(chrom_exon_df):
    """
    Genes in the order the reference emits them (sorted by gene_end, first appearance: reads_coverage_merge.py:264-267)
    and, per gene, the merged 0-based half-open intervals covering the union of its exons (the positions
    `np.unique(flatten(arange(start - 1, end)))` of :331-349).
    """
    df = chrom_exon_df.sort_values('gene_end', axis=0)
    genes = df['gene'].unique().tolist()
    # one pass over the table (round 3 filtered the DataFrame once per gene: O(genes x exons) on a real chromosome)
    codes = {g: k for k, g in enumerate(genes)}
    per_gene = [[] for _ in genes]
    for g, a, b in zip(df['gene'].tolist(), (df['start'].values - 1).tolist(), df['end'].values.tolist()):
        per_gene[codes[g]].append((a, b))
    out = []
    for iv in per_gene:
        iv.sort()
        merged = []
        for a, b in iv:
            if b <= a:
                continue
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        out.append(merged)
    return genes, out


def _load_sample_rows(data_dir, sample_ids, chrom):
    """CSR rows of every sample (None where the file is missing) and the chromosome length."""
    from scipy import sparse
    rows, n = [], None
    for s in sample_ids:
        f = os.path.join(data_dir, s, 'chrom_coverage_{0}_{1}.npz'.format(s, chrom))
        if os.path.isfile(f):
            m = sparse.load_npz(f).tocsr()
            if m.shape[0] != 1:
                m = m.transpose().tocsr()
            m.sum_duplicates()
            rows.append(m)
            n = m.shape[1] if n is None else max(n, m.shape[1])
        else:
            rows.append(None)
    return rows, n


def assemble_chrom_packed(data_dir, sample_ids, chrom_exon_df, device=None, verbose=True, chunk=4096):
    """
    :return: (genes, packed float32 1-d array, lengths int64, device milliseconds) or (None, None, None, 0.0) when no
             sample has coverage for the chromosome (the reference returns an empty dict there, :247-252).
    """
    unique_chrom = chrom_exon_df.chr.unique()
    if len(unique_chrom) > 1:
        raise ValueError('chrom_exon_df contains exon data for more than one chromosome!')
    chrom = unique_chrom[0]
    rows, n = _load_sample_rows(data_dir, sample_ids, chrom)
    if n is None:
        if verbose:
            logging.info('CHR {0} -- no chromosome coverage files available.'.format(chrom))
        return None, None, None, 0.0
    for s, r in zip(sample_ids, rows):
        if r is None and verbose:
            logging.info('CHR {0} -- nonexistent chromosome coverage file for {1} (imputing zeroes).'.format(chrom, s))

    genes, ivs = gene_intervals(chrom_exon_df)
    lengths = np.array([sum(b - a for a, b in iv) for iv in ivs], dtype=np.int64)
    c_gene, c_src, c_dst, c_len = [], [], [], []
    for g, iv in enumerate(ivs):
        col = 0
        for a, b in iv:
            for s0 in range(a, b, chunk):
                ln = min(chunk, b - s0)
                c_gene.append(g); c_src.append(s0); c_dst.append(col); c_len.append(ln)
                col += ln
    c_gene = np.asarray(c_gene, dtype=np.int32)
    c_src = np.asarray(c_src, dtype=np.int64)
    c_dst = np.asarray(c_dst, dtype=np.int64)
    c_len = np.asarray(c_len, dtype=np.int32)

    p = len(sample_ids)
    nnz = np.zeros(p, dtype=np.int64)
    keep_i, keep_v = [], []
    inexact = 0
    for i, r in enumerate(rows):
        if r is None:
            keep_i.append(np.zeros(0, dtype=np.int32)); keep_v.append(np.zeros(0, dtype=np.float32))
            continue
        idx = np.ascontiguousarray(r.indices, dtype=np.int32)
        val64 = np.asarray(r.data, dtype=np.float64)
        val = np.ascontiguousarray(val64, dtype=np.float32)
        inexact += int(np.count_nonzero(val.astype(np.float64) != val64))
        nnz[i] = idx.size
        keep_i.append(idx); keep_v.append(val)
    if inexact:
        logging.warning('{0} coverage values are not exactly representable in float32.'.format(inexact))

    lib = _lib.load()
    iptrs = (ctypes.c_void_p * p)(*[a.ctypes.data if a.size else None for a in keep_i])
    vptrs = (ctypes.c_void_p * p)(*[a.ctypes.data if a.size else None for a in keep_v])
    packed = np.empty(int((lengths * p).sum()), dtype=np.float32)
    ms = ctypes.c_double(0.0)
    dev = int(os.environ.get('LOCAL_RANK', 0)) if device is None else int(device)
    rc = lib.dn_assemble_coverage(dev, int(n), p, _lib._p(nnz, ctypes.c_int64), iptrs, vptrs, len(genes),
                                  _lib._p(lengths, ctypes.c_int64), int(c_gene.size), _lib._p(c_gene, ctypes.c_int32),
                                  _lib._p(c_src, ctypes.c_int64), _lib._p(c_dst, ctypes.c_int64),
                                  _lib._p(c_len, ctypes.c_int32), _lib._p(packed, ctypes.c_float), ctypes.byref(ms))
    if rc != 0:
        raise _lib.DegnormAmdError('dn_assemble_coverage failed ({0}): {1}'.format(
            rc, lib.dn_assemble_last_error().decode('utf-8', 'replace')))
    return genes, packed, lengths, float(ms.value)


def merge_chrom_coverage(data_dir, sample_ids, chrom_exon_df, verbose=True, device=None):
    """
    Drop-in for the reference's `merge_chrom_coverage` (reads_coverage_merge.py:167-372): {gene: (p x L) float64}
    with genes in the reference's insertion order; an empty dict when no sample has the chromosome.
    """
    genes, packed, lengths, _ = assemble_chrom_packed(data_dir, sample_ids, chrom_exon_df, device=device, verbose=verbose)
    if genes is None:
        return dict()
    p = len(sample_ids)
    out, o = dict(), 0
    for g, L in zip(genes, lengths):
        cnt = p * int(L)
        out[g] = packed[o:o + cnt].reshape(p, int(L)).astype(np.float64)
        o += cnt
    if verbose:
        logging.info('CHR {0} -- obtained {1} coverage matrices.'.format(chrom_exon_df.chr.unique()[0], len(out)))
    return out
