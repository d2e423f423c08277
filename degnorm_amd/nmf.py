"""
GeneNMFOA -- host-side mirror of the reference's NMF over-approximation driver (degnorm/nmf.py:10-711)
whose arithmetic runs in hand-written HIP kernels on an MI355X (libdegnorm_amd.so).

Same constructor, ``run(cov_dat, reads_dat)`` (plus the ``fit`` alias BASELINE.json asks for),
``save_results(...)`` and post-run attributes as the reference, so ``from degnorm_amd.nmf import *`` can
replace ``from degnorm.nmf import *`` in degnorm/__main__.py:11.

``run()`` is a single-rank ``ShardedNMFOA`` (nmf_mpi.py of this package, ``LocalComm``): the engine bench.py measures.
On the device: ratio_svd (nmf.py:109-121) and the initial normalisation (:524-535), adjust_coverage_curves (:142-146,
folded into the loads), baseline_selection with its nmf() calls (:189-372) for every gene, the DI clip (:398-399),
correct_di_scores (:148-158) and the read-count / scale-factor update (:575-590).  The host packs the coverage dict once
(threads, float64 -> float32 staging, dn_upload_ragged), sees 3p + 4 numbers per outer iteration, and copies rho / x_adj /
x_weighted / ran_baseline_selection back once at the end (fetch_state) plus the estimates of the last iteration.

There is no CPU fallback: if the HIP library or a GPU is missing, run() raises.
"""
import logging
import os
import time
import warnings

import numpy as np

__all__ = ['GeneNMFOA']


class GeneNMFOA(object):

    def __init__(self, degnorm_iter=5, downsample_rate=1, min_high_coverage=50,
                 nmf_iter=100, bins=20, n_jobs=1, skip_baseline_selection=False, random_state=123,
                 device=None):
        """
        Same parameters as the reference constructor (nmf.py:12-53).  ``n_jobs`` is accepted for
        compatibility: a value > 1 is the number of host packing threads, the default 1 (the reference's
        --proc-per-node default) lets the library use the box's cores, at most 16; ``device`` (extra) picks the
        HIP device (default: LOCAL_RANK or 0).
        """
        self.degnorm_iter = abs(int(degnorm_iter))
        self.nmf_iter = abs(int(nmf_iter))
        self.n_jobs = abs(int(n_jobs))
        self.bins = abs(int(bins))
        self.min_high_coverage = max(2, abs(int(min_high_coverage)))
        self.min_bins = np.ceil(self.bins * 0.2)
        self.downsample_rate = abs(int(downsample_rate))
        self.mem_splits = None
        self.x = None
        self.x_weighted = None
        self.x_adj = None
        self.p = None
        self.n_genes = None
        self.genes = None
        self.norm_factors = None
        self.scale_factors = None
        self.rho = None
        self.fitted = False
        self.ran_baseline_selection = None
        self.skip_baseline_selection = skip_baseline_selection
        self.random_state = random_state
        # svds(k=1) needs >= 2 columns, so downsampled runs only ask for 2 (nmf.py:51-53)
        if self.downsample_rate > 1:
            self.min_high_coverage = 2

        self.device = int(os.environ.get('LOCAL_RANK', 0)) if device is None else int(device)
        self.downsample_offsets = None     # optional (degnorm_iter x n_genes) explicit start offsets
        self.solver_step_cap = None        # optional: step cap of one on-chip eigen-solve (default 4000)
        self.traces = []                   # per outer iteration: (n_genes x TRACE_LEN) int32 device traces
        self.kernel_ms = []                # per outer iteration: device time of the main kernel
        self._dev = None
        self._engine = None                # the single-rank ShardedNMFOA behind run()
        self.timings = None                # wall time per stage of the last run()

    # ------------------------------------------------------------------------------------------- #
    def check_input(self, cov_mats):
        """The reference's input checks (nmf.py:455-481), same exception types and messages."""
        if self.x.shape[0] != self.n_genes:
            raise ValueError('Number of genes in read count matrix not equal to number of coverage matrices!')
        if not all(getattr(z, 'ndim', 0) == 2 for z in cov_mats):
            raise ValueError('Not all coverage matrices are 2-d arrays!')
        li_vec = np.array([z.shape[1] for z in cov_mats])
        if np.sum(li_vec / self.p < 1) > 0:
            logging.warning('At least one coverage matrix is taller than it is wide.'
                            'Ensure that coverage matrices are shaped (p x L_i).')
        if self.downsample_rate > 1:
            if not np.min(li_vec) >= self.downsample_rate:
                raise ValueError('downsample_rate is too large; take-every size > at least one gene.')

    # ------------------------------------------------------------------------------------------- #
    def run(self, cov_dat, reads_dat):
        """
        Run the DegNorm iterations (reference: GeneNMFOA.run, nmf.py:483-601).

        :param cov_dat: OrderedDict {gene: (p x L_g) coverage matrix}
        :param reads_dat: (n_genes x p) read counts
        :return: list of (p x L_g) float64 estimated coverage matrices, in gene order

        ``self.timings`` (extra) holds the wall time of each stage: ``pack_upload_s`` (coverage dict -> float32 staging ->
        HBM, read counts), ``run_s`` (initial pass + the outer iterations), ``estimates_s`` (rebuild + D2H of the last
        iteration's estimates), ``fetch_state_s`` (D2H of rho / x_adj / x_weighted / flags).
        """
        from .nmf_mpi import ShardedNMFOA, LocalComm
        self.n_genes = len(cov_dat)
        self.genes = list(cov_dat.keys())
        self.x = np.array(reads_dat, dtype=np.float64)
        cov_mats = list(cov_dat.values())
        self.p = cov_mats[0].shape[0]
        self.ran_baseline_selection = np.zeros(shape=[self.n_genes, self.degnorm_iter]).astype(bool)
        self.check_input(cov_mats)

        t0 = time.perf_counter()
        eng = ShardedNMFOA(comm=LocalComm(), device=self.device, degnorm_iter=self.degnorm_iter,
                           downsample_rate=self.downsample_rate, min_high_coverage=self.min_high_coverage,
                           nmf_iter=self.nmf_iter, bins=self.bins, skip_baseline_selection=self.skip_baseline_selection,
                           random_state=self.random_state)
        eng.gene_names = self.genes                       # failures and unconverged solves are reported by name
        eng.downsample_offsets = self.downsample_offsets  # optional explicit systematic-sample starts
        if self.degnorm_iter == 0:
            eng.device_outer = False                      # nothing to iterate: the initial DI scores come back as they are
        self._engine = eng
        self._dev = eng.dev                               # raises here when the HIP library or a GPU is missing
        if self.solver_step_cap is not None:
            eng.dev.set_solver_step_cap(self.solver_step_cap)
        eng.load(cov_mats, self.x, n_threads=self.n_jobs if self.n_jobs > 1 else 0)
        if eng.dev.inexact:
            warnings.warn('{0} coverage values are not exactly representable in float32; '
                          'they were rounded on upload.'.format(eng.dev.inexact))
        t1 = time.perf_counter()

        eng.initialize()                                  # nmf.py:521-535
        logging.info('Initial sequencing depth scale factors -- \n\t{0}'
                     .format(', '.join([str(x) for x in eng.scale_factors])))
        estimates, t_est = None, 0.0
        for i in range(self.degnorm_iter):                # nmf.py:558-596
            last = i == self.degnorm_iter - 1
            eng.iterate(i, want_estimates=last)
            if not self.skip_baseline_selection:
                logging.info('DegNorm iteration {0} -- {1} genes sent through baseline selection'
                             .format(i + 1, eng.n_flagged[-1]))
            logging.info('DegNorm iteration {0} -- sequencing depth scale factors: \n\t{1}'
                         .format(i + 1, ', '.join([str(x) for x in eng.scale_factors])))
            if last:
                te = time.perf_counter()
                estimates = eng.dev.fetch_estimates()     # built with this iteration's scale factors (nmf.py:601)
                t_est = time.perf_counter() - te
        t2 = time.perf_counter()
        eng.fetch_state()
        t3 = time.perf_counter()

        self.rho, self.x_adj, self.x_weighted = eng.rho, eng.x_adj, eng.x_weighted
        self.norm_factors, self.scale_factors = eng.norm_factors, eng.scale_factors
        if self.degnorm_iter > 0:
            self.ran_baseline_selection = np.asarray(eng.ran_baseline_selection, dtype=bool)
        self.traces, self.kernel_ms = eng.traces, eng.kernel_ms
        self.timings = {'pack_upload_s': t1 - t0, 'run_s': t2 - t1 - t_est, 'estimates_s': t_est, 'fetch_state_s': t3 - t2}
        self.fitted = True
        if estimates is None:                                       # degnorm_iter == 0: ratio-SVD estimates are not kept
            estimates = [np.array(c, dtype=np.float64) for c in cov_mats]
        return estimates

    fit = run   # BASELINE.json names the entry point `fit`; the reference's is `run`

    def correct_di_scores(self):
        """
        The reference's public helper (nmf.py:148-158), kept for callers and subclasses that invoke it: genes whose DI row is
        all zero get 1 - colsum(x_weighted) / colsum(x_adj).  run() does this on the device (dn_outer_apply); here the same
        rule is applied to the host copies self.rho / self.x_weighted / self.x_adj, in place.
        """
        if not self.fitted:
            raise ValueError('Model not yet fit. NMF-OA has not been run.')
        avg_di_score = 1 - (self.x_weighted.sum(axis=0) / self.x_adj.sum(axis=0))
        zero_idx = np.where(self.rho.max(axis=1) == 0)[0]
        if len(zero_idx) > 0:
            self.rho[zero_idx, :] = avg_di_score

    def estimates_for(self, genes):
        """
        Estimated coverage matrices of a few genes only (by name), rebuilt on demand from the device-side state of the
        last iteration -- what the plot / report steps need (`__main__.py:291-316`, `report.py:97-113`) without
        materialising all n estimates.
        """
        if not self.fitted or self._dev is None:
            raise ValueError('Model not yet fit. NMF-OA has not been run.')
        pos = {g: k for k, g in enumerate(self.genes)}
        return self._dev.fetch_estimates_subset([pos[g] for g in genes])

    # ------------------------------------------------------------------------------------------- #
    def save_results(self, estimates, gene_manifest_df, output_dir='.', sample_ids=None):
        """
        Write the reference's result files (nmf.py:603-711): per-chromosome
        ``<chr>/estimated_coverage_matrices_<chr>.pkl`` plus ``degradation_index_scores.csv``,
        ``adjusted_read_counts.csv`` and ``ran_baseline_selection.csv`` (columns chr, gene, then samples /
        iter_k), rows in ``self.genes`` order.
        """
        from .results import write_results
        if not self.fitted:
            raise ValueError('Model not yet fit. NMF-OA has not been run.')
        write_results(genes=self.genes, estimates=estimates, rho=self.rho, x_adj=self.x_adj,
                      ran_baseline_selection=self.ran_baseline_selection, gene_manifest_df=gene_manifest_df,
                      output_dir=output_dir, sample_ids=sample_ids, p=self.p, degnorm_iter=self.degnorm_iter)
