"""
SURVEY 8(f-1)/(f-2), BASELINE config 5 shape: warm-start directory -> CLI gene filter -> NMF-OA core -> result files.
Golden: tests/golden/warm.npz, produced by the reference's load_from_previous + GeneNMFOA.run + save_results on the same
synthetic directory (tests/golden/make_golden.py::sec_warm).  The loader/filter test runs on CPU; the end-to-end
CSV parity test needs the GPU.
"""
import os
import pickle

import numpy as np
import pytest

from conftest import golden
from degnorm_amd import synth  # noqa: F401
import _fixtures
from degnorm_amd.warm_start import load_from_previous, select_genes, run_from_warm_start


def _make_dir(tmp_path, G):
    src = str(tmp_path / 'prev_run')
    sample_ids = _fixtures.write_warm_start_dir(src, seed=int(G['seed']), n_genes=int(G['n_genes']), p=int(G['p']),
                                            l_min=int(G['l_min']), l_max=int(G['l_max']))
    assert sample_ids == list(G['sample_ids'])
    return src


def test_loader_and_gene_filter_match_reference(tmp_path):
    G = golden('warm')
    src = _make_dir(tmp_path, G)
    new = tmp_path / 'new_run'
    new.mkdir()
    dat = load_from_previous(src, str(new))
    # gene order = per-chromosome pickle order restricted to genes known to both tables (warm_start.py:59-97)
    assert list(dat['gene_cov_dict'].keys()) == list(G['loaded_order'])
    assert list(dat['genes_df'].gene) == list(G['loaded_order']) == list(dat['read_count_df'].gene)
    assert 'ORPHAN_PKL' not in dat['gene_cov_dict'] and dat['sample_ids'] == list(G['sample_ids'])
    assert (new / 'read_counts.csv').exists() and (new / 'chr2' / 'coverage_matrices_chr2.pkl').exists()
    cov, reads_df, genes_df = select_genes(dat['gene_cov_dict'], dat['read_count_df'], dat['genes_df'],
                                           minimax_coverage=int(G['minimax']))
    assert list(cov.keys()) == list(G['genes']) == list(genes_df.gene) == list(reads_df.gene)
    assert list(genes_df.chr) == list(G['chr'])
    with pytest.raises(ValueError, match='No genes available'):
        d2 = load_from_previous(src)
        select_genes(d2['gene_cov_dict'], d2['read_count_df'], d2['genes_df'], minimax_coverage=10 ** 9)
    with pytest.raises(IOError):
        load_from_previous(src, str(tmp_path / 'does_not_exist'))
    # MPI-only limits (__main_mpi__.py:374-376)
    d3 = load_from_previous(src)
    first = next(iter(d3['gene_cov_dict']))
    d3['gene_cov_dict'][first] = d3['gene_cov_dict'][first].copy()
    d3['gene_cov_dict'][first][0, 0] = 2.0 ** 31
    kept, _, _ = select_genes(d3['gene_cov_dict'], d3['read_count_df'], d3['genes_df'], mpi_limits=True)
    assert first not in kept


@pytest.mark.gpu
def test_warm_start_end_to_end_csv_parity(tmp_path):
    """degradation_index_scores.csv / adjusted_read_counts.csv / ran_baseline_selection.csv vs the reference's."""
    import pandas as pd
    G = golden('warm')
    src = _make_dir(tmp_path, G)
    out = tmp_path / 'out'
    out.mkdir()
    run_from_warm_start(src, str(out), degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']),
                        minimax_coverage=int(G['minimax']))
    sid = list(G['sample_ids'])
    di = pd.read_csv(out / 'degradation_index_scores.csv')
    adj = pd.read_csv(out / 'adjusted_read_counts.csv')
    ran = pd.read_csv(out / 'ran_baseline_selection.csv')
    assert list(di.columns) == ['chr', 'gene'] + sid and list(di.gene) == list(G['genes']) and list(di.chr) == list(G['chr'])
    np.testing.assert_allclose(di[sid].values, G['rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(adj[sid].values, G['x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(ran[['iter_0', 'iter_1']].values.astype(bool), G['ran'])
    with open(out / 'chr1' / 'estimated_coverage_matrices_chr1.pkl', 'rb') as f:
        e1 = pickle.load(f)
    assert list(e1.keys()) == list(G['chr1_genes'])
    np.testing.assert_allclose(np.vstack([v.sum(axis=1) for v in e1.values()]), G['chr1_est_rowsum'], rtol=1e-9)


def test_sidecar_gives_the_same_inputs_and_goes_stale(tmp_path):
    """The packed float32 side-car: same genes, same order, same values as the pickles; ignored once a pickle changes."""
    from degnorm_amd.warm_start import write_sidecars
    G = golden('warm')
    src = _make_dir(tmp_path, G)
    ref = load_from_previous(src, use_sidecar=False)
    inexact = write_sidecars(src)
    assert sorted(inexact) == ['chr1', 'chr2', 'chrX'] and not any(inexact.values())
    dat = load_from_previous(src)
    assert list(dat['gene_cov_dict'].keys()) == list(ref['gene_cov_dict'].keys()) == list(G['loaded_order'])
    for g, m in dat['gene_cov_dict'].items():
        assert m.dtype == np.float32 and np.array_equal(m, ref['gene_cov_dict'][g])
    assert dat['read_count_df'].equals(ref['read_count_df']) and dat['genes_df'].equals(ref['genes_df'])
    # the CLI filter works on the views
    cov, _, genes_df = select_genes(dat['gene_cov_dict'], dat['read_count_df'], dat['genes_df'], minimax_coverage=int(G['minimax']))
    assert list(cov.keys()) == list(G['genes'])
    # a rewritten pickle makes the side-car stale
    pk = os.path.join(src, 'chr1', 'coverage_matrices_chr1.pkl')
    with open(pk, 'rb') as f:
        d = pickle.load(f)
    first = next(iter(d))
    d[first] = d[first] + 1.0
    with open(pk, 'wb') as f:
        pickle.dump(d, f)
    again = load_from_previous(src)
    assert again['gene_cov_dict'][first].dtype == np.float64 and np.array_equal(again['gene_cov_dict'][first], d[first])
    other = next(g for g in again['gene_cov_dict'] if g not in d)          # another chromosome still comes from its side-car
    assert again['gene_cov_dict'][other].dtype == np.float32
    # a truncated / foreign data file that still passes the stat check of its index: fall back to the pickle, do not crash
    npy = os.path.join(src, 'chr2', 'coverage_matrices_chr2.f32.npy')
    full = np.load(npy)
    np.save(npy, full[:-7])
    again = load_from_previous(src)
    chr2 = [g for g in ref['gene_cov_dict'] if g in again['gene_cov_dict'] and again['gene_cov_dict'][g].dtype == np.float64 and g not in d]
    assert chr2 and all(np.array_equal(again['gene_cov_dict'][g], ref['gene_cov_dict'][g]) for g in chr2)
    with open(npy, 'wb') as f:
        f.write(b'not a numpy file')
    again = load_from_previous(src)
    assert all(np.array_equal(again['gene_cov_dict'][g], ref['gene_cov_dict'][g]) for g in chr2)


def _sharded_warm_start(tmp_path, G, size, use_sidecar):
    """run_from_warm_start_mpi on `size` ranks (threads, bare send/recv communicator); returns the output directory."""
    import threading
    from test_sharded_cpu import _ThreadComm
    from degnorm_amd.warm_start import run_from_warm_start_mpi, write_sidecars
    src = _make_dir(tmp_path, G)
    if use_sidecar:
        write_sidecars(src)
    out = tmp_path / 'out_mpi'
    out.mkdir()
    comm, errs = _ThreadComm(size), [None] * size

    def work(r):
        try:
            run_from_warm_start_mpi(comm.view(r), src, str(out), degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']),
                                    minimax_coverage=int(G['minimax']))
        except Exception as e:          # noqa: BLE001
            errs[r] = e
    ths = [threading.Thread(target=work, args=(r,)) for r in range(size)]
    [t.start() for t in ths]
    [t.join(timeout=600) for t in ths]
    assert not any(t.is_alive() for t in ths) and errs == [None] * size, errs
    return out


def _check_csvs(out, G):
    import pandas as pd
    sid = list(G['sample_ids'])
    di = pd.read_csv(out / 'degradation_index_scores.csv')
    adj = pd.read_csv(out / 'adjusted_read_counts.csv')
    ran = pd.read_csv(out / 'ran_baseline_selection.csv')
    assert list(di.columns) == ['chr', 'gene'] + sid and list(di.gene) == list(G['genes']) and list(di.chr) == list(G['chr'])
    np.testing.assert_allclose(di[sid].values, G['rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(adj[sid].values, G['x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(ran[['iter_0', 'iter_1']].values.astype(bool), G['ran'])
    with open(out / 'chr1' / 'estimated_coverage_matrices_chr1.pkl', 'rb') as f:
        e1 = pickle.load(f)
    assert list(e1.keys()) == list(G['chr1_genes'])
    np.testing.assert_allclose(np.vstack([v.sum(axis=1) for v in e1.values()]), G['chr1_est_rowsum'], rtol=1e-9)


@pytest.mark.parametrize('use_sidecar', [False, True])
def test_sharded_warm_start_cpu(tmp_path, monkeypatch, oracle, use_sidecar):
    """
    BASELINE config 5's shape on the sharded driver (`__main_mpi__.py:357-456`): warm-start directory -> MPI gene filter ->
    3 ranks -> result files == the reference's.  Host logic only (the oracle stands in for the device), with and
    without the float32 side-car.
    """
    from _oracle_device import OracleDevice
    monkeypatch.setattr('degnorm_amd._lib.Device', OracleDevice)
    G = golden('warm')
    _check_csvs(_sharded_warm_start(tmp_path, G, 3, use_sidecar), G)


def test_sharded_warm_start_errors_reach_every_rank(tmp_path, monkeypatch, oracle):
    import threading
    from test_sharded_cpu import _ThreadComm
    from degnorm_amd.warm_start import run_from_warm_start_mpi
    from _oracle_device import OracleDevice
    monkeypatch.setattr('degnorm_amd._lib.Device', OracleDevice)
    G = golden('warm')
    src = _make_dir(tmp_path, G)
    out = tmp_path / 'o'
    out.mkdir()
    comm, errs = _ThreadComm(2), [None, None]

    def work(r):
        try:
            run_from_warm_start_mpi(comm.view(r), src, str(out), minimax_coverage=10 ** 9)
        except Exception as e:          # noqa: BLE001
            errs[r] = e
    ths = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in ths]
    [t.join(timeout=120) for t in ths]
    assert not any(t.is_alive() for t in ths)
    assert all(isinstance(e, ValueError) and 'No genes available' in str(e) for e in errs)


@pytest.mark.gpu
def test_sharded_warm_start_on_the_gpu(tmp_path):
    """The same chain with the kernels: two ranks (threads, both on device 0), float32 side-car, CSV parity with the reference."""
    G = golden('warm')
    _check_csvs(_sharded_warm_start(tmp_path, G, 2, True), G)
