"""
Pins the CPU oracle (oracle/nmfoa_oracle.c + oracle/oracle.py) to golden vectors produced by the REAL
reference (tests/golden/make_golden.py imports /root/reference/degnorm/nmf.py and nmf_mpi.py).
Runs on the CPU (-m "not gpu").  Agreement is at float64 round-off: the oracle is a restatement, not an
approximation.
"""
import numpy as np
import pytest

from conftest import golden, input_checksum
from degnorm_amd import synth

RT = 1e-9


def _genes(seed, gene_ids, p, l_min, l_max):
    return [synth.synth_gene(int(seed), int(g), int(p), int(l_min), int(l_max))[0] for g in gene_ids]


def _fixture_inputs(G):
    covs = _genes(G['seed'], G['gene_ids'], G['p'], G['l_min'], G['l_max'])
    cks = np.array([input_checksum(c) for c in covs])
    np.testing.assert_array_equal(cks, G['checksum'], err_msg='synthetic generator drifted from the fixture inputs')
    return covs


def test_kat_nmf_rank_one_ratio_svd(oracle):
    """G1/G2: nmf() (nmf.py:78-107), rank_one_approx (:55-64), ratio_svd (:109-121)."""
    K = golden('kat')
    for k in range(int(K['n_nmf'])):
        x = K['nmf%d_x' % k]
        Kk, Ek = oracle.nmf(x, int(K['nmf%d_T' % k]))
        np.testing.assert_allclose(Kk.dot(Ek), K['nmf%d_KE' % k], rtol=RT, atol=1e-9)
        np.testing.assert_allclose(np.abs(Kk).ravel(), K['nmf%d_absK' % k], rtol=RT, atol=1e-9)
        K1, E1 = oracle.rank_one(x)
        np.testing.assert_allclose(K1.dot(E1), K['nmf%d_r1KE' % k], rtol=RT, atol=1e-9)
        np.testing.assert_allclose(oracle.ratio_svd(x), K['nmf%d_ratio' % k], rtol=RT, atol=1e-9)


def test_kat_split_into_chunks_and_shift_bins(oracle):
    """utils.split_into_chunks (utils.py:176-192) incl. the 'fewer than n chunks' case; shift_bins (nmf.py:160-187)."""
    from degnorm_amd.utils import split_into_chunks, chunk_bounds
    K = golden('kat')
    for i, (ln, nb) in enumerate(K['chunk_cases']):
        want = K['chunk%d_lens' % i]
        got = [len(c) for c in oracle.split_into_chunks(int(ln), int(nb))]
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal([len(c) for c in split_into_chunks(list(range(ln)), int(nb))], want)
        np.testing.assert_array_equal(np.diff(chunk_bounds(int(ln), int(nb))), want)
    # bins stay contiguous after drops: the oracle / device represent them as lengths only
    lens = [len(c) for c in oracle.split_into_chunks(41, 5)]
    for d, key in zip([1, 0, 2], ['shift_bounds0', 'shift_bounds1', 'shift_bounds2']):
        del lens[d]
        np.testing.assert_array_equal(np.concatenate([[0], np.cumsum(lens)]), K[key])


def test_baseline_selection_genes(oracle):
    """G3: per-gene baseline_selection over all classes, with call traces; skip_baseline_selection variant."""
    G = golden('genes')
    covs = _fixture_inputs(G)
    prm = oracle.make_params(nmf_iter=int(G['nmf_iter']))
    rho, flags, trace, est = oracle.baseline_batch(covs, G['scale'], prm, want_estimates=True)
    np.testing.assert_array_equal(flags, G['flags'])
    np.testing.assert_array_equal(trace[:, 1], G['n_calls'])
    np.testing.assert_array_equal(trace[:, 2], G['sum_cols'])
    np.testing.assert_array_equal(trace[G['n_calls'] > 0, 0], G['n0'][G['n_calls'] > 0])
    np.testing.assert_allclose(rho, G['rho'], rtol=RT, atol=1e-12)
    for k, e in enumerate(est):
        np.testing.assert_allclose(e.sum(axis=1), G['est_rowsum'][k], rtol=RT)
        step = max(1, e.shape[1] // 16)
        np.testing.assert_allclose(e[:, ::step][:, :16], G['est_sample'][k], rtol=RT, atol=1e-9)
    # every exit code is exercised by the fixture
    assert set(trace[:, 3]) >= {0, 1, 3, 4, 6}
    prm_s = oracle.make_params(nmf_iter=int(G['nmf_iter']), skip_baseline_selection=True)
    rho_s, flags_s, _, _ = oracle.baseline_batch(covs, G['scale'], prm_s)
    assert not flags_s.any()
    np.testing.assert_allclose(rho_s, G['rho_skip'], rtol=RT, atol=1e-12)


def _check_run(oracle, name, rt=RT):
    G = golden(name)
    covs = _fixture_inputs(G)
    hist = {}
    ds = G['offsets'] if 'offsets' in G.files else None
    out = oracle.run(covs, G['reads'], degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']),
                     downsample_rate=int(G['downsample_rate']), ds_starts=ds, want_estimates=True, history=hist)
    for i in range(int(G['degnorm_iter'])):
        np.testing.assert_allclose(hist['rho'][i] if False else hist['rho'][i], hist['rho'][i])
        np.testing.assert_array_equal(hist['trace'][i][:, 1], G['n_calls'][i])
        np.testing.assert_array_equal(hist['trace'][i][:, 2], G['sum_cols'][i])
    np.testing.assert_array_equal(out['ran_baseline_selection'], G['ran_baseline_selection'])
    np.testing.assert_allclose(out['rho'], G['rho'], rtol=rt, atol=1e-12)
    np.testing.assert_allclose(out['x_adj'], G['x_adj'], rtol=rt)
    np.testing.assert_allclose(out['scale_factors'], G['scale_factors'], rtol=rt)
    np.testing.assert_allclose(out['x_weighted'], G['x_weighted'], rtol=rt)
    np.testing.assert_allclose(np.vstack([e.sum(axis=1) for e in out['estimates']]), G['est_rowsum'], rtol=rt)
    k = 0
    while 'est_%d' % k in G.files:
        np.testing.assert_allclose(out['estimates'][k], G['est_%d' % k], rtol=rt, atol=1e-9)
        k += 1
    return G, out


def test_run_config1(oracle):
    """G4: GeneNMFOA.run on config 1 (100 genes x 4 samples x L=1000, 1 iteration), nmf.py:483-601."""
    _check_run(oracle, 'run_c1')


def test_run_config2_subset(oracle):
    """G4: 64-gene draw of config 2 (p=10, L~U[200,5000]), 3 outer iterations."""
    _check_run(oracle, 'run_c2')


def test_run_config2_deep(oracle):
    """
    G4 at depth: 256 genes of config 2 (ids disjoint from run_c2), BASELINE's 5 outer iterations, T = 100 -- both device
    gene classes (L <= / > ~2047) and every exit of baseline_selection; per-iteration call counts and active columns exact.
    """
    G, out = _check_run(oracle, 'run_c2_deep')
    assert int(G['degnorm_iter']) == 5 and len(G['gene_ids']) == 256
    rel = np.abs(out['rho'] - G['rho']) / np.maximum(np.abs(G['rho']), 1e-300)
    assert rel.max() < 1e-5                                           # BASELINE.json's stated DI tolerance


def test_run_downsampled(oracle):
    """G6: take-every 50 (p=6) and the config-4 regime (p=50, take-every 500, active matrices 50 x <=10)."""
    _check_run(oracle, 'run_dsamp50')
    _check_run(oracle, 'run_dsamp500')


def test_mpi_twin_matches_single_node(oracle):
    """G5: run_gene_nmfoa_mpi (nmf_mpi.py:555-863) through a fake communicator == GeneNMFOA.run == oracle."""
    G = golden('mpi')
    covs = _fixture_inputs(G)
    out = oracle.run(covs, G['reads'], degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']))
    for tag in ('single', 'mpi2', 'mpi3'):
        np.testing.assert_allclose(out['rho'], G[tag + '_rho'], rtol=RT, atol=1e-12)
        np.testing.assert_allclose(out['x_adj'], G[tag + '_x_adj'], rtol=RT)
        np.testing.assert_array_equal(out['ran_baseline_selection'], G[tag + '_flags'])
    from degnorm_amd.utils import split_into_chunks
    for size in (2, 3):
        np.testing.assert_array_equal([len(c) for c in split_into_chunks(list(range(len(covs))), size)],
                                      G['mpi%d_chunk_lens' % size])


def sparse_golden_cases():
    """tests/golden/sparse.npz (make_golden.py sec_sparse): per gene the inputs, the parameters and the reference's outputs."""
    G = golden('sparse')
    for k in range(int(G['n'])):
        T, bins, mhc, rate, off, flag, decoupled = [int(v) for v in G['prm%d' % k]]
        yield dict(k=k, x=G['x%d' % k].astype(np.float64), scale=G['scale%d' % k], T=T, bins=bins, mhc=mhc, rate=rate, off=off,
                   flag=bool(flag), decoupled=bool(decoupled), rho=G['rho%d' % k], calls=G['calls%d' % k],
                   est_rowsum=G['est_rowsum%d' % k], kind=str(G['kinds'][k]))


def test_sparse_genes_vs_reference_golden(oracle):
    """
    G3b: sparse genes run through the reference -- samples in blocks without a common base (the top singular vector jumps between
    the blocks as lambda grows: nmf.py:88-99 with a solver that must find the TOP triplet every time), samples that lose all
    coverage when a bin is dropped (nmf.py:315, exact-zero row sums of K E), fewer active columns than bins or samples.
    """
    n = n_dec = n_zero = 0
    for c in sparse_golden_cases():
        prm = oracle.make_params(nmf_iter=c['T'], bins=c['bins'], min_high_coverage=c['mhc'], downsample_rate=c['rate'])
        kw = {} if c['rate'] == 1 else {'ds_start': np.array([c['off']], dtype=np.int64)}
        rho, flags, trace, est = oracle.baseline_batch([c['x']], c['scale'], prm, want_estimates=True, **kw)
        msg = 'sparse golden gene %d (%s)' % (c['k'], c['kind'])
        assert trace[0, 1] == len(c['calls']) and trace[0, 2] == c['calls'].sum(), msg
        assert bool(flags[0]) == c['flag'], msg
        np.testing.assert_allclose(rho[0], c['rho'], rtol=1e-8, atol=1e-10, err_msg=msg)
        np.testing.assert_allclose(est[0].sum(axis=1), c['est_rowsum'], rtol=1e-8, atol=1e-8, err_msg=msg)
        n += 1; n_dec += c['decoupled']; n_zero += int(trace[0, 4] == 3)
    assert n >= 60 and n_dec >= 20 and n_zero >= 3          # the fixture holds what it was made for


def pileup_golden_cases():
    """tests/golden/pileup.npz (make_golden.py pileup): read pile-up genes with the reference's outputs (three agreeing runs each)."""
    from degnorm_amd import synth
    G = golden('pileup')
    for k in range(int(G['n'])):
        gene, p, T, kind, flag = [int(v) for v in G['prm%d' % k]]
        x, kind_g = synth.pileup_gene(int(G['seed']), gene, p, int(G['l_min']), int(G['l_max']))
        assert kind_g == kind and input_checksum(x) == float(G['ck%d' % k])              # the generator has not drifted
        yield dict(k=k, x=x, p=p, T=T, kind=kind, flag=bool(flag), scale=G['scale%d' % k], rho=G['rho%d' % k], calls=G['calls%d' % k],
                   est_rowsum=G['est_rowsum%d' % k])


def pileup_run_inputs(G):
    from degnorm_amd import synth
    cov_dat, reads, kinds = synth.pileup_dataset(int(G['run_seed']), G['run_gene_ids'], int(G['run_p']), int(G['run_l_min']), int(G['run_l_max']))
    covs = list(cov_dat.values())
    np.testing.assert_array_equal([input_checksum(c) for c in covs], G['run_checksum'])
    np.testing.assert_array_equal(reads, G['run_reads'])
    return covs, reads


def test_pileup_genes_vs_reference_golden(oracle):
    """
    G3c (round 4): the input kind DegNorm really sees -- reads stacked into piecewise-constant small-integer coverage
    (reads.py:714,773) -- where 10 x == max ties and equal bin means are everywhere.  120 genes (p = 4 / 6 / 10, T = 20 / 100),
    each run three times through the reference's baseline_selection (all 120 stable): call sequence, flag, DI, estimate row sums.
    """
    n = n_loop = 0
    for c in pileup_golden_cases():
        rho, flags, trace, est = oracle.baseline_batch([c['x']], c['scale'], oracle.make_params(nmf_iter=c['T']), want_estimates=True)
        msg = 'pileup golden gene %d (kind %d, p = %d, T = %d)' % (c['k'], c['kind'], c['p'], c['T'])
        assert trace[0, 1] == len(c['calls']) and trace[0, 2] == c['calls'].sum(), msg
        assert bool(flags[0]) == c['flag'], msg
        np.testing.assert_allclose(rho[0], c['rho'], rtol=1e-8, atol=1e-10, err_msg=msg)
        np.testing.assert_allclose(est[0].sum(axis=1), c['est_rowsum'], rtol=1e-8, atol=1e-8, err_msg=msg)
        n += 1; n_loop += int(c['flag'])
    assert n >= 100 and n_loop >= 40


def test_pileup_run_vs_reference_golden(oracle):
    """The whole chain on pile-up coverage: GeneNMFOA.run on 48 genes (p = 6), 3 outer iterations, against the reference's run."""
    G = golden('pileup')
    covs, reads = pileup_run_inputs(G)
    hist = {}
    out = oracle.run(covs, reads, degnorm_iter=3, nmf_iter=100, history=hist)
    for i in range(3):
        np.testing.assert_array_equal(hist['trace'][i][:, 1], G['run_n_calls'][i])
        np.testing.assert_array_equal(hist['trace'][i][:, 2], G['run_sum_cols'][i])
    np.testing.assert_array_equal(out['ran_baseline_selection'], G['run_flags'])
    np.testing.assert_allclose(out['rho'], G['run_rho'], rtol=RT, atol=1e-12)
    np.testing.assert_allclose(out['x_adj'], G['run_x_adj'], rtol=RT)
    np.testing.assert_allclose(out['scale_factors'], G['run_scale_factors'], rtol=RT)


def steps_golden_cases():
    """tests/golden/steps.npz (make_golden.py steps): the fuzz generator's `steps` kind, genes stable over three runs of the reference."""
    G = golden('steps')
    for k in range(int(G['n'])):
        T, bins, mhc, flag = [int(v) for v in G['prm%d' % k]]
        yield dict(k=k, x=G['x%d' % k].astype(np.float64), scale=G['scale%d' % k], T=T, bins=bins, mhc=mhc, flag=bool(flag),
                   rho=G['rho%d' % k], calls=G['calls%d' % k], est_rowsum=G['est_rowsum%d' % k])


def steps_agreement(run_gene):
    """(genes, DI mismatches, bin-sequence mismatches) of `run_gene(case) -> (rho, flag, trace row, estimate)` against the golden."""
    n, bad_di, bad_seq = 0, [], []
    for c in steps_golden_cases():
        rho, flag, tr, est = run_gene(c)
        di_ok = (tr[1] == len(c['calls']) and bool(flag) == c['flag'] and np.allclose(rho, c['rho'], rtol=1e-8, atol=1e-10)
                 and np.allclose(est.sum(axis=1), c['est_rowsum'], rtol=1e-8, atol=1e-8))
        if not di_ok:
            bad_di.append(c['k'])
        elif tr[2] != c['calls'].sum():
            bad_seq.append(c['k'])                        # same DI and estimate through another bin of an exact tie
        n += 1
    return n, bad_di, bad_seq


def test_steps_kind_vs_reference_golden(oracle):
    """
    G3d (round 4): piecewise-constant small-integer coverage with empty stretches -- exact ties between bin means and exact-zero
    residuals -- is the one input kind on which round 3's randomised runs left device / oracle disagreements unexplained by the
    reference.  Measured with the reference: 30 of 300 such genes do not agree with THEMSELVES over three runs (ARPACK's random start
    decides a tie); on the 254 that do, the oracle must reproduce the reference's DI, flag, call count and estimate on all but a
    handful (a tie between bin means decided by summation order: numpy sums pairwise, the oracle sequentially), and may take another
    bin of an exact tie with the SAME DI on a few more.
    """
    def run(c):
        rho, flags, trace, est = oracle.baseline_batch([c['x']], c['scale'], oracle.make_params(nmf_iter=c['T'], bins=c['bins'], min_high_coverage=c['mhc']),
                                                       want_estimates=True)
        return rho[0], flags[0], trace[0], est[0]
    G = golden('steps')
    assert int(G['tried']) == 300 and int(G['unstable']) >= 20          # the reference's own noise floor on this kind is ~10 %
    n, bad_di, bad_seq = steps_agreement(run)
    print('steps kind, oracle vs reference: %d genes, DI / flag / calls differ on %s, same DI through another bin on %s' % (n, bad_di, bad_seq))
    assert n >= 250 and len(bad_di) <= 3 and len(bad_seq) <= 8
