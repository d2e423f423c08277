"""
GeneNMFOA -- host-side mirror of the reference's NMF over-approximation driver (degnorm/nmf.py:10-711)
whose per-gene arithmetic runs in hand-written HIP kernels on an MI355X (libdegnorm_amd.so).

Same constructor, ``run(cov_dat, reads_dat)`` (plus the ``fit`` alias BASELINE.json asks for),
``save_results(...)`` and post-run attributes as the reference, so ``from degnorm_amd.nmf import *`` can
replace ``from degnorm.nmf import *`` in degnorm/__main__.py:11.

What stays on the host (float64 numpy, O(n_genes x p) per outer iteration): the DI clip (nmf.py:398-399),
correct_di_scores (nmf.py:148-158) and the read-count / scale-factor update (nmf.py:575-590).
What runs on the device: ratio_svd (nmf.py:109-121), adjust_coverage_curves (nmf.py:142-146, folded into
the loads) and baseline_selection with its nmf() calls (nmf.py:189-372) for every gene.

There is no CPU fallback: if the HIP library or a GPU is missing, run() raises.
"""
import logging
import os
import pickle as pkl
import warnings

import numpy as np

from . import _lib

__all__ = ['GeneNMFOA']


class GeneNMFOA(object):

    def __init__(self, degnorm_iter=5, downsample_rate=1, min_high_coverage=50,
                 nmf_iter=100, bins=20, n_jobs=1, skip_baseline_selection=False, random_state=123,
                 device=None):
        """
        Same parameters as the reference constructor (nmf.py:12-53).  ``n_jobs`` is accepted for
        compatibility and only sizes the host packing threads; ``device`` (extra) picks the HIP device
        (default: LOCAL_RANK or 0).
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

    def _offsets_for_iteration(self, i, rng):
        """Systematic-sample start per gene (nmf.py:422).  Explicit offsets win (parity tests)."""
        if self.downsample_rate <= 1:
            return None
        if self.downsample_offsets is not None:
            return np.asarray(self.downsample_offsets[i], dtype=np.int64)
        return rng.randint(0, self.downsample_rate, size=self.n_genes).astype(np.int64)

    def correct_di_scores(self):
        """Genes that were not run through baseline selection get the sample-average DI (nmf.py:148-158)."""
        untouched = self.rho.max(axis=1) == 0
        if np.sum(untouched) > 0:
            self.rho[untouched, :] = 1 - (self.x_weighted.sum(axis=0) / self.x_adj.sum(axis=0))

    # ------------------------------------------------------------------------------------------- #
    def run(self, cov_dat, reads_dat):
        """
        Run the DegNorm iterations (reference: GeneNMFOA.run, nmf.py:483-601).

        :param cov_dat: OrderedDict {gene: (p x L_g) coverage matrix}
        :param reads_dat: (n_genes x p) read counts
        :return: list of (p x L_g) float64 estimated coverage matrices, in gene order
        """
        self.n_genes = len(cov_dat)
        self.genes = list(cov_dat.keys())
        self.x = np.array(reads_dat, dtype=np.float64)
        cov_mats = list(cov_dat.values())
        self.p = cov_mats[0].shape[0]
        self.ran_baseline_selection = np.zeros(shape=[self.n_genes, self.degnorm_iter]).astype(bool)
        self.check_input(cov_mats)

        dev = _lib.Device(self.device)
        self._dev = dev
        dev.hint_downsample(self.downsample_rate)
        if self.solver_step_cap is not None:
            dev.set_solver_step_cap(self.solver_step_cap)
        dev.upload(cov_mats, n_threads=max(self.n_jobs, 0))
        if dev.inexact:
            warnings.warn('{0} coverage values are not exactly representable in float32; '
                          'they were rounded on upload.'.format(dev.inexact))

        # ---- initialisation (nmf.py:521-535) ----
        est_sums, cov_sums, status = dev.ratio_svd_sums()
        if np.any(status == -4):
            raise ValueError('the rank-1 SVD did not converge within the step cap on {0} gene(s) during initialisation '
                             '(ARPACK would raise ArpackNoConvergence): first {1}'
                             .format(int(np.sum(status == -4)), self.genes[int(np.argmax(status == -4))]))
        if np.any(status != 0):
            raise ValueError('rank-1 SVD failed on {0} gene(s) during initialisation (all-zero coverage?): first {1}'
                             .format(int(np.sum(status != 0)), self.genes[int(np.argmax(status != 0))]))
        self.rho = 1 - (cov_sums / (est_sums + 1))
        low_di_gene = self.rho.max(axis=1) < 0.1
        count_sums = self.x[low_di_gene, :].sum(axis=0) if np.any(low_di_gene) else self.x.sum(axis=0)
        self.norm_factors = count_sums / np.median(count_sums)
        self.x_weighted = self.x / self.norm_factors
        self.scale_factors = np.copy(self.norm_factors)
        logging.info('Initial sequencing depth scale factors -- \n\t{0}'
                     .format(', '.join([str(x) for x in self.scale_factors])))

        rng = np.random.RandomState(self.random_state)
        self.traces, self.kernel_ms = [], []
        estimates = None

        # ---- DegNorm iterations (nmf.py:558-596) ----
        for i in range(self.degnorm_iter):
            last = i == self.degnorm_iter - 1
            rho, flags, trace = dev.baseline_iteration(
                self.scale_factors, nmf_iter=self.nmf_iter, bins=self.bins, min_high_coverage=self.min_high_coverage,
                downsample_rate=self.downsample_rate, skip_baseline_selection=self.skip_baseline_selection,
                want_estimates=last, ds_start=self._offsets_for_iteration(i, rng))
            self.traces.append(trace)
            self.kernel_ms.append(dev.last_kernel_ms())
            bad = trace[:, 6] != 0
            if np.any(bad):
                logging.warning('DegNorm iteration {0} -- {1} gene(s) hit a degenerate factorization '
                                '(reference would raise); their DI scores were left at 0.'.format(i + 1, int(bad.sum())))
            noconv = np.flatnonzero(trace[:, 6] == -4)
            if noconv.size:
                logging.warning('DegNorm iteration {0} -- the rank-1 SVD did not converge within the step cap on {1} gene(s) '
                                '(ARPACK would raise ArpackNoConvergence); not used, DI left at 0: {2}{3}'
                                .format(i + 1, noconv.size, ', '.join(self.genes[k] for k in noconv[:10]),
                                        ' ...' if noconv.size > 10 else ''))

            rho[rho > 0.9] = 0.9                                    # nmf.py:398
            rho[rho < 0.] = 0.                                      # nmf.py:399
            self.rho = rho
            self.ran_baseline_selection[:, i] = flags               # nmf.py:403
            if not self.skip_baseline_selection:
                logging.info('DegNorm iteration {0} -- {1} genes sent through baseline selection'
                             .format(i + 1, np.sum(self.ran_baseline_selection[:, i])))

            self.x_adj = self.x_weighted / (1 - self.rho)           # nmf.py:575
            self.correct_di_scores()                                # nmf.py:578
            self.x_adj = self.x_weighted / (1 - self.rho)           # nmf.py:581
            col = self.x_adj.sum(axis=0)
            self.norm_factors = col / np.median(col)                # nmf.py:584
            self.x_weighted = self.x_weighted / self.norm_factors   # nmf.py:587
            if last:
                estimates = dev.fetch_estimates()                   # built with this iteration's scale factors
            self.scale_factors = self.scale_factors * self.norm_factors   # nmf.py:590
            logging.info('DegNorm iteration {0} -- sequencing depth scale factors: \n\t{1}'
                         .format(i + 1, ', '.join([str(x) for x in self.scale_factors])))

        self.fitted = True
        if estimates is None:                                       # degnorm_iter == 0: ratio-SVD estimates are not kept
            estimates = [np.array(c, dtype=np.float64) for c in cov_mats]
        return estimates

    fit = run   # BASELINE.json names the entry point `fit`; the reference's is `run`

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
