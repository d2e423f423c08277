"""
Generate golden vectors for the NMF-OA hot path by running the REAL reference implementation
(/root/reference/degnorm/nmf.py, nmf_mpi.py) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]

The reference is imported read-only and never travels: only the .npz outputs written next to this
script are committed.  Inputs are either stored inline (small KATs) or re-generated from
degnorm_amd.synth by (seed, gene id) -- each fixture then carries a per-gene input checksum so a
drifting generator is detected instead of silently compared.

Sections:
    kat        G1/G2  nmf() / ratio_svd() / rank_one_approx() / split_into_chunks / shift_bins KATs
    genes      G3     per-gene baseline_selection() outputs + call traces over all gene classes
    run_c1     G4     GeneNMFOA.run on config 1 (100 x 4 x 1000, 1 iteration)
    run_c2     G4     GeneNMFOA.run on a 64-gene draw of config 2 (p=10, L~U[200,5000]), 3 iterations
    run_c2_deep G4    the same on 256 other genes for BASELINE's 5 iterations (both device gene classes, every exit)
    mpi        G5     run_gene_nmfoa_mpi through an in-process fake communicator (2 and 3 ranks)
    dsamp      G6     downsampled runs (rate 50) with captured systematic-sample offsets
    warm       G7     warm-start directory -> filter -> run -> save_results CSVs
    merge      f-3    merge_chrom_coverage on per-sample chromosome CSR vectors
    sparse     G3b    baseline_selection on sparse genes: decoupled sample blocks, samples losing all coverage (three stable runs each)
    steps      G3d    baseline_selection on the fuzz generator's `steps` kind (piecewise-constant small integers with empty stretches: exact ties
                      between bin means, exact-zero residuals), three runs each; how many are unstable in the reference itself is recorded
    pileup     G3c    baseline_selection on read pile-up coverage (piecewise-constant small integers: DegNorm's real input kind), three runs
                      each, + GeneNMFOA.run on 48 such genes for 3 iterations
"""
import os
import sys
import time
import threading
import queue
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from degnorm.nmf import GeneNMFOA            # noqa: E402  (the reference)
import degnorm.nmf_mpi as ref_mpi            # noqa: E402
from degnorm.utils import split_into_chunks  # noqa: E402
from degnorm_amd import synth                # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _fixtures                             # noqa: E402


def checksum(cov):
    """Order-sensitive input checksum of one coverage matrix (float64 exact for integer counts)."""
    w = (np.arange(cov.size, dtype=np.float64) % 251.) + 1.
    return float((cov.reshape(-1) * w).sum())


class Tracer:
    """Wraps a GeneNMFOA instance: records per-gene nmf() call sizes and downsample offsets."""

    def __init__(self, model):
        self.model = model
        self.calls = []          # list (per baseline_selection call) of lists of n columns
        self.offsets = []        # systematic-sample starts in call order
        self.rho_hist = []       # post-clip rho per outer iteration
        self.scale_hist = []     # scale factors used by each outer iteration
        self._nmf = model.nmf
        self._bs = model.baseline_selection
        self._pabs = model.par_apply_baseline_selection
        self._adj = model.adjust_coverage_curves
        model.nmf = self.nmf
        model.baseline_selection = self.baseline_selection
        model.par_apply_baseline_selection = self.pabs
        model.adjust_coverage_curves = self.adj
        tracer = self

        def sys_sample(n, take_every):
            out = GeneNMFOA._systematic_sample(n, take_every)
            tracer.offsets.append(int(out[0]) if not np.isscalar(out) else int(out))
            return out
        model._systematic_sample = sys_sample

    def nmf(self, x, factors=False):
        out = self._nmf(x, factors=factors)      # a call that raises (svds ValueError, nmf.py:306-310) is not counted
        self.calls[-1].append(x.shape[1])
        return out

    def baseline_selection(self, F):
        self.calls.append([])
        return self._bs(F)

    def pabs(self, dat, degnorm_iter):
        out = self._pabs(dat, degnorm_iter)
        self.rho_hist.append(self.model.rho.copy())
        return out

    def adj(self, dat):
        self.scale_hist.append(self.model.scale_factors.copy())
        return self._adj(dat)


def trace_arrays(calls):
    n_calls = np.array([len(c) for c in calls], dtype=np.int32)
    sum_cols = np.array([sum(c) for c in calls], dtype=np.int64)
    n0 = np.array([c[0] if c else -1 for c in calls], dtype=np.int32)
    return n_calls, sum_cols, n0


# -------------------------------------------------------------------------------------------- #
def sec_kat():
    rng = np.random.default_rng(11)
    out = {}
    m = GeneNMFOA(nmf_iter=100)
    k = 0
    for (p, n, T) in [(2, 2, 20), (4, 64, 20), (4, 64, 100), (10, 1000, 100), (3, 7, 50), (6, 5, 30)]:
        K0 = rng.lognormal(0, 0.5, size=(p, 1))
        E0 = 20 + 50 * np.abs(np.sin(np.linspace(0, 3, n)))[None, :]
        x = rng.poisson(K0 * E0 * np.linspace(0.4, 1, n)[None, :] ** rng.integers(0, 2, size=(p, 1))).astype(float)
        m.nmf_iter = T
        K, E = m.nmf(x, factors=True)
        K1, E1 = m.rank_one_approx(x)
        out['nmf{0}_x'.format(k)] = x
        out['nmf{0}_T'.format(k)] = T
        out['nmf{0}_KE'.format(k)] = K.dot(E)
        out['nmf{0}_absK'.format(k)] = np.abs(K).ravel()
        out['nmf{0}_r1KE'.format(k)] = K1.dot(E1)
        out['nmf{0}_ratio'.format(k)] = m.ratio_svd(x)
        k += 1
    out['n_nmf'] = k
    # split_into_chunks / shift_bins (utils.py:176-192, nmf.py:160-187)
    cases = [(201, 20), (1000, 20), (50, 20), (19, 20), (20, 20), (2600, 20), (7, 3), (399, 20), (4999, 20)]
    out['chunk_cases'] = np.array(cases)
    for i, (ln, nb) in enumerate(cases):
        ch = split_into_chunks(list(range(ln)), nb)
        out['chunk{0}_lens'.format(i)] = np.array([len(c) for c in ch])
    bins = split_into_chunks(list(range(41)), 5)
    seq = []
    for d in [1, 0, 2]:
        del bins[d]
        bins = GeneNMFOA.shift_bins(bins, d)
        seq.append(np.array([b[0] for b in bins] + [bins[-1][-1] + 1]))
    out['shift_bounds0'], out['shift_bounds1'], out['shift_bounds2'] = seq
    np.savez_compressed(os.path.join(HERE, 'kat.npz'), **out)
    print('kat done')


def sec_genes():
    """Per-gene baseline_selection over every class; scale factors fixed, p=6, T=100, L<=1500 to bound time."""
    seed, p = 3, 6
    l_min, l_max = 200, 1500
    gene_ids = list(range(72))
    scale = np.array([0.8, 1.1, 0.95, 1.3, 1.0, 0.9])
    m = GeneNMFOA(nmf_iter=100)
    m.p = p
    tr = Tracer(m)
    rho, flags, cks, cls_all, est_rs, est_samp, Ls = [], [], [], [], [], [], []
    t0 = time.time()
    for g in gene_ids:
        cov, cls = synth.synth_gene(seed, g, p, l_min, l_max)
        F = (cov.T / scale).T
        r, est, fl = m.baseline_selection(F)
        rho.append(r); flags.append(fl); cks.append(checksum(cov)); cls_all.append(cls)
        est_rs.append(est.sum(axis=1)); Ls.append(cov.shape[1])
        est_samp.append(est[:, :: max(1, cov.shape[1] // 16)][:, :16])
    n_calls, sum_cols, n0 = trace_arrays(tr.calls)
    # skip_baseline_selection variant on the same genes
    m2 = GeneNMFOA(nmf_iter=100, skip_baseline_selection=True)
    m2.p = p
    rho_skip = []
    for g in gene_ids:
        cov, _ = synth.synth_gene(seed, g, p, l_min, l_max)
        rho_skip.append(m2.baseline_selection((cov.T / scale).T)[0])
    np.savez_compressed(os.path.join(HERE, 'genes.npz'), seed=seed, p=p, l_min=l_min, l_max=l_max,
                        gene_ids=np.array(gene_ids), scale=scale, rho=np.vstack(rho), flags=np.array(flags),
                        checksum=np.array(cks), classes=np.array(cls_all), n_calls=n_calls, sum_cols=sum_cols,
                        n0=n0, est_rowsum=np.vstack(est_rs), est_sample=np.stack(est_samp), L=np.array(Ls),
                        rho_skip=np.vstack(rho_skip), nmf_iter=100)
    print('genes done in %.1fs' % (time.time() - t0), 'calls hist', np.bincount(n_calls))


def _run_and_save(name, seed, n_total, p, l_min, l_max, gene_ids, degnorm_iter, nmf_iter, downsample_rate=1,
                  keep_est=4):
    cov_dat, reads, classes = synth.synth_dataset(seed, n_total, p, l_min, l_max, gene_ids=gene_ids)
    m = GeneNMFOA(degnorm_iter=degnorm_iter, nmf_iter=nmf_iter, downsample_rate=downsample_rate, n_jobs=1)
    tr = Tracer(m)
    t0 = time.time()
    est = m.run(cov_dat, reads)
    dt = time.time() - t0
    n = len(gene_ids)
    n_calls, sum_cols, n0 = trace_arrays(tr.calls)
    out = dict(seed=seed, p=p, l_min=l_min, l_max=l_max, gene_ids=np.array(gene_ids), degnorm_iter=degnorm_iter,
               nmf_iter=nmf_iter, downsample_rate=downsample_rate, reads=reads, classes=classes,
               checksum=np.array([checksum(c) for c in cov_dat.values()]),
               rho=m.rho, x_adj=m.x_adj, ran_baseline_selection=m.ran_baseline_selection,
               scale_factors=m.scale_factors, norm_factors=m.norm_factors, x_weighted=m.x_weighted,
               rho_hist=np.stack(tr.rho_hist), scale_hist=np.stack(tr.scale_hist),
               n_calls=n_calls.reshape(degnorm_iter, n), sum_cols=sum_cols.reshape(degnorm_iter, n),
               n0=n0.reshape(degnorm_iter, n), ref_seconds=dt,
               est_rowsum=np.vstack([e.sum(axis=1) for e in est]))
    if downsample_rate > 1:
        out['offsets'] = np.array(tr.offsets).reshape(degnorm_iter, n)
    for k in range(min(keep_est, n)):
        out['est_{0}'.format(k)] = est[k]
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%s done in %.1fs (%.2f gene-iters/s)' % (name, dt, n * degnorm_iter / dt))


def sec_run_c1():
    c = synth.CONFIGS['c1']
    _run_and_save('run_c1', c['seed'], c['n_genes'], c['p'], c['l_min'], c['l_max'], list(range(c['n_genes'])),
                  degnorm_iter=1, nmf_iter=100)


def sec_run_c2():
    c = synth.CONFIGS['c2']
    _run_and_save('run_c2', c['seed'], c['n_genes'], c['p'], c['l_min'], c['l_max'], list(range(64)),
                  degnorm_iter=3, nmf_iter=100)


def sec_run_c2_deep():
    # the headline configuration at depth: 256 genes (ids disjoint from run_c2), BASELINE's 5 outer iterations, T = 100;
    # both gene classes of the device (L <= / > ~2047) and every exit of baseline_selection occur (~25 min of reference time)
    c = synth.CONFIGS['c2']
    _run_and_save('run_c2_deep', c['seed'], c['n_genes'], c['p'], c['l_min'], c['l_max'], list(range(64, 320)),
                  degnorm_iter=5, nmf_iter=100, keep_est=2)


def sec_dsamp():
    # rate 50 on p=6 genes (L >= 200 > rate); offsets captured in call order (iteration-major, gene-minor).
    _run_and_save('run_dsamp50', 6, 48, 6, 200, 3000, list(range(48)), degnorm_iter=2, nmf_iter=50,
                  downsample_rate=50)
    # config-4 regime: p=50, rate 500, L in 501..5000 (active matrices 50 x <=10, n < p)
    _run_and_save('run_dsamp500', 4, 24, 50, 501, 5000, list(range(24)), degnorm_iter=2, nmf_iter=100,
                  downsample_rate=500, keep_est=2)


class FakeComm:
    """In-process stand-in for an mpi4py communicator (.size .rank .send .recv .Barrier), one per thread."""

    def __init__(self, size):
        self.size = size
        self.boxes = {}
        self.lock = threading.Lock()
        self.barrier = threading.Barrier(size)

    def view(self, rank):
        parent = self

        class View:
            size = parent.size

            def __init__(self):
                self.rank = rank

            def _box(self, src, dst, tag):
                with parent.lock:
                    return parent.boxes.setdefault((src, dst, tag), queue.Queue())

            def send(self, obj, dest, tag=0):
                self._box(self.rank, dest, tag).put(obj)

            def recv(self, source, tag=0):
                return self._box(source, self.rank, tag).get()

            def Barrier(self):
                parent.barrier.wait()
        return View()


def _sparse_gene(rng, p, L, kind):
    """Sparse / low-count coverage whose samples fall apart into blocks without a common base, or lose all coverage as bins drop."""
    env = 1.0 + np.abs(np.sin(np.linspace(0, rng.uniform(0.5, 7), L) + rng.uniform(0, 3)))
    if kind == 'steps':                                    # piecewise constant small integers, many zero stretches
        k = int(rng.integers(1, 9))
        edges = np.sort(rng.integers(0, L, size=k))
        lvl = rng.integers(0, 6, size=(p, k + 1)).astype(float)
        x = np.zeros((p, L))
        prev = 0
        for j, e in enumerate(list(edges) + [L]):
            x[:, prev:e] = lvl[:, j:j + 1]
            prev = e
        return x
    depth = float(rng.choice([0.15, 0.3, 1.0, 3.0]))
    mean = depth * np.outer(rng.lognormal(0, 0.5, p), env)
    if kind == 'decay':
        for i in range(p):
            if rng.random() < 0.5:
                mean[i] *= np.linspace(rng.uniform(0.02, 0.8), 1.0, L) ** rng.uniform(0.5, 3)
    x = rng.poisson(mean).astype(float)
    if kind == 'holes':
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(1, max(2, L // 3))))
            x[:, a:b] = 0.0
    if kind == 'blocks':                                   # two groups of samples covering different halves of the transcript
        h = int(rng.integers(1, p)); cut = int(rng.integers(L // 4, 3 * L // 4))
        x[:h, cut:] = 0.0
        x[h:, :cut] = 0.0
    return x


def sec_sparse():
    """
    baseline_selection on sparse genes (added in round 3 after a randomised device-vs-oracle run): samples in blocks
    without a common base (the top singular vector jumps between blocks as lambda grows), samples that lose all coverage when a
    bin is dropped (nmf.py:315), fewer active columns than bins / than samples.  The reference's ARPACK start is random: every
    gene is run three times and kept only when the call sequence, the flag and rho (1e-9) agree between the runs.
    """
    rng = np.random.default_rng(2026)
    keep = []
    tried = 0
    t0 = time.time()
    while len(keep) < 80 and tried < 2500:
        tried += 1
        p = int(rng.choice([2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 33, 50]))
        rate = int(rng.choice([1, 1, 40, 200]))
        L = int(rng.integers(rate + 1, 13 * rate + 1)) if rate > 1 else int(rng.integers(20, 700))
        L = min(L, 1800)
        kind = str(rng.choice(['steps', 'holes', 'decay', 'plain', 'blocks', 'blocks']))
        x = _sparse_gene(rng, p, L, kind)
        scale = np.exp(rng.uniform(-0.5, 0.5, p)) if rng.random() < 0.5 else np.linspace(0.9, 1.15, p)
        T = int(rng.choice([1, 5, 20, 40]))
        bins = int(rng.choice([2, 5, 20, 20]))
        mhc = 2 if rate > 1 else int(rng.choice([2, 10]))
        off = int(rng.integers(0, rate)) if rate > 1 else 0
        F = (x.T / scale).T
        runs = []
        for rep in range(3):
            m = GeneNMFOA(degnorm_iter=1, nmf_iter=T, downsample_rate=rate, bins=bins, n_jobs=1)
            m.min_high_coverage = mhc
            m.p = p
            calls = []
            inner = m.nmf

            def nmf(xx, factors=False, inner=inner, calls=calls):
                out = inner(xx, factors=factors)
                calls.append(xx.shape[1])
                if factors:
                    # nmf.py:315 tests min row sum of K E == 0.  ARPACK returns an EXACT zero for a sample without coverage in the
                    # remaining columns; for a sample that is merely decoupled from the top block it returns round-off that is
                    # sometimes exactly zero and sometimes 1e-16 (measured: 53 exact / 250 garbage / 97 mixed of 400 random block
                    # matrices) -- there the reference's branch is a coin flip, and such genes are not golden material
                    ke0 = np.min(np.abs(out[0]).dot(np.abs(out[1])).sum(axis=1)) == 0
                    if ke0 != (np.min(xx.sum(axis=1)) == 0):
                        calls.append(-1)
                return out
            m.nmf = nmf
            m._systematic_sample = lambda n, take_every, off=off: np.arange(off, n, take_every)
            try:
                r, est, fl = m.baseline_selection(F.copy())
            except Exception:
                runs = None
                break
            runs.append((np.asarray(r, dtype=float).ravel(), bool(fl), list(calls), est.sum(axis=1)))
        if not runs:
            continue
        stable = all(rr[1] == runs[0][1] and rr[2] == runs[0][2] and np.allclose(rr[0], runs[0][0], rtol=1e-9, atol=1e-11)
                     and np.allclose(rr[3], runs[0][3], rtol=1e-8, atol=1e-8) for rr in runs[1:])
        if not stable or len(runs[0][2]) == 0 or any(-1 in rr[2] for rr in runs):
            continue
        # interesting = decoupled supports among the first call's active columns, or a short call sequence
        cols = np.arange(off, L, rate) if rate > 1 else np.arange(L)
        act = [j for j in cols if F[:, j].max() > 0.1 * F.max()]
        A = F[:, act] > 0
        reach = np.zeros(p, bool); reach[0] = True
        for _ in range(p):
            hit = A[:, (A & reach[:, None]).any(axis=0)].any(axis=1)
            if (hit | reach).sum() == reach.sum():
                break
            reach |= hit
        decoupled = not reach.all()
        if not decoupled and rng.random() < 0.75:
            continue
        keep.append(dict(x=x.astype(np.float32), scale=scale, T=T, bins=bins, mhc=mhc, rate=rate, off=off, kind=kind,
                         rho=runs[0][0], flag=runs[0][1], calls=runs[0][2], est_rowsum=runs[0][3], decoupled=decoupled))
    out = dict(n=len(keep))
    for k, g in enumerate(keep):
        out['x%d' % k] = g['x']; out['scale%d' % k] = g['scale']; out['rho%d' % k] = g['rho']; out['est_rowsum%d' % k] = g['est_rowsum']
        out['calls%d' % k] = np.array(g['calls'], dtype=np.int32)
        out['prm%d' % k] = np.array([g['T'], g['bins'], g['mhc'], g['rate'], g['off'], int(g['flag']), int(g['decoupled'])], dtype=np.int64)
    out['kinds'] = np.array([g['kind'] for g in keep])
    np.savez_compressed(os.path.join(HERE, 'sparse.npz'), **out)
    print('sparse: kept %d of %d tried (%d with decoupled samples) in %.0f s' % (len(keep), tried, sum(g['decoupled'] for g in keep), time.time() - t0))


def sec_steps():
    """
    The `steps` kind of tools/fuzz_parity.py (round 3: the one kind on which device and oracle still disagreed on 0.6-0.9 % of the
    genes, explained by summation order -- exact ties between bin means -- but never checked against the reference): piecewise-
    constant coverage from a handful of levels 0..5 with empty stretches.  300 such genes (p = 2 .. 16, T = 5 / 20 / 40, bins 5 / 20,
    min_high_coverage 2 / 10), three runs each through the reference.  Kept: the genes whose three runs agree (call sequence, flag,
    DI to 1e-9, estimate row sums); recorded: how many do NOT agree with themselves -- the reference's own noise floor on this kind.
    """
    rng = np.random.default_rng(777)
    keep, unstable, raised, tried = [], 0, 0, 0
    t0 = time.time()
    while tried < 300:
        tried += 1
        p = int(rng.choice([2, 3, 4, 6, 8, 10, 12, 16]))
        L = int(rng.integers(40, 900))
        x = _sparse_gene(rng, p, L, 'steps')
        if not x.any():
            continue
        scale = np.exp(rng.uniform(-0.5, 0.5, p)) if rng.random() < 0.5 else np.linspace(0.9, 1.15, p)
        T = int(rng.choice([5, 20, 40]))
        bins = int(rng.choice([5, 20]))
        mhc = int(rng.choice([2, 10]))
        F = (x.T / scale).T
        runs = []
        for rep in range(3):
            m = GeneNMFOA(degnorm_iter=1, nmf_iter=T, bins=bins, n_jobs=1)
            m.min_high_coverage = mhc
            m.p = p
            calls = []
            inner = m.nmf

            def nmf(xx, factors=False, inner=inner, calls=calls):
                out = inner(xx, factors=factors)
                calls.append(xx.shape[1])
                if factors:                                                # the nmf.py:315 coin flip (see sec_sparse)
                    ke0 = np.min(np.abs(out[0]).dot(np.abs(out[1])).sum(axis=1)) == 0
                    if ke0 != (np.min(xx.sum(axis=1)) == 0):
                        calls.append(-1)
                return out
            m.nmf = nmf
            try:
                r, est, fl = m.baseline_selection(F.copy())
            except Exception:
                runs = None
                break
            runs.append((np.asarray(r, dtype=float).ravel(), bool(fl), list(calls), est.sum(axis=1)))
        if not runs:
            raised += 1
            continue
        stable = all(rr[1] == runs[0][1] and rr[2] == runs[0][2] and np.allclose(rr[0], runs[0][0], rtol=1e-9, atol=1e-11)
                     and np.allclose(rr[3], runs[0][3], rtol=1e-8, atol=1e-8) for rr in runs[1:])
        if not stable or any(-1 in rr[2] for rr in runs):
            unstable += 1
            continue
        if len(runs[0][2]) == 0 and rng.random() < 0.8:                   # most genes of this kind leave before the first nmf(): keep a few
            continue
        keep.append(dict(x=x.astype(np.float32), scale=scale, T=T, bins=bins, mhc=mhc, rho=runs[0][0], flag=runs[0][1],
                         calls=runs[0][2], est_rowsum=runs[0][3]))
    out = dict(n=len(keep), tried=tried, unstable=unstable, raised=raised)
    for k, g in enumerate(keep):
        out['x%d' % k] = g['x']; out['scale%d' % k] = g['scale']; out['rho%d' % k] = g['rho']; out['est_rowsum%d' % k] = g['est_rowsum']
        out['calls%d' % k] = np.array(g['calls'], dtype=np.int32)
        out['prm%d' % k] = np.array([g['T'], g['bins'], g['mhc'], int(g['flag'])], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'steps.npz'), **out)
    print('steps: %d tried, %d raised in the reference, %d unstable or coin-flip in the reference itself, %d kept (%d through the drop loop), %.0f s'
          % (tried, raised, unstable, len(keep), sum(len(g['calls']) > 1 for g in keep), time.time() - t0))


def sec_pileup():
    """
    Read pile-up coverage (synth.pileup_gene: reads of 75-150 bases stacked at low / medium depth with 3' bias -> piecewise-constant
    small integers, what reads.py:714,773 really produces) through the reference.  Such genes are full of exact ties (10 x == max in
    get_high_coverage_idx, equal bin means in the drop loop, exact-zero residuals), i.e. the inputs on which summation order can
    decide a branch.  Every gene is run THREE times through baseline_selection (ARPACK's start vector is random) and kept when call
    sequence, flag, rho (1e-9) and estimate row sums agree; the number of unstable genes is recorded too -- that is the reference's
    own noise floor on this input kind.  p in {4, 6, 10}, T in {20, 100}; then one GeneNMFOA.run (p = 6, 48 genes, 3 iterations).
    """
    rng = np.random.default_rng(4242)
    keep, unstable, tried = [], 0, 0
    t0 = time.time()
    g = 0
    while len(keep) < 120 and tried < 400:
        tried += 1
        p = int(rng.choice([4, 6, 10]))
        T = int(rng.choice([20, 100]))
        x, kind = synth.pileup_gene(11, g, p, 300, 2200)
        g += 1
        scale = np.exp(rng.uniform(-0.4, 0.4, p)) if rng.random() < 0.7 else np.ones(p)
        F = (x.T / scale).T
        runs = []
        for rep in range(3):
            m = GeneNMFOA(degnorm_iter=1, nmf_iter=T, n_jobs=1)
            m.p = p
            calls = []
            inner = m.nmf

            def nmf(xx, factors=False, inner=inner, calls=calls):
                out = inner(xx, factors=factors)
                calls.append(xx.shape[1])
                return out
            m.nmf = nmf
            try:
                r, est, fl = m.baseline_selection(F.copy())
            except Exception:
                runs = None
                break
            runs.append((np.asarray(r, dtype=float).ravel(), bool(fl), list(calls), est.sum(axis=1)))
        if not runs:
            continue
        stable = all(rr[1] == runs[0][1] and rr[2] == runs[0][2] and np.allclose(rr[0], runs[0][0], rtol=1e-9, atol=1e-11)
                     and np.allclose(rr[3], runs[0][3], rtol=1e-8, atol=1e-8) for rr in runs[1:])
        if not stable:
            unstable += 1
            continue
        keep.append(dict(gene=g - 1, p=p, T=T, kind=kind, scale=scale, rho=runs[0][0], flag=runs[0][1], calls=runs[0][2],
                         est_rowsum=runs[0][3], ck=checksum(x)))
    out = dict(n=len(keep), seed=11, l_min=300, l_max=2200, tried=tried, unstable=unstable)
    for k, q in enumerate(keep):
        out['scale%d' % k] = q['scale']; out['rho%d' % k] = q['rho']; out['est_rowsum%d' % k] = q['est_rowsum']
        out['calls%d' % k] = np.array(q['calls'], dtype=np.int32)
        out['prm%d' % k] = np.array([q['gene'], q['p'], q['T'], q['kind'], int(q['flag'])], dtype=np.int64)
        out['ck%d' % k] = q['ck']
    print('pileup genes: kept %d of %d tried, %d unstable in the reference itself, %.0f s' % (len(keep), tried, unstable, time.time() - t0))
    # whole chain on the same input kind: 48 genes, p = 6, 3 outer iterations (scale factors, DI, flags, traces per iteration)
    gene_ids = list(range(1000, 1048))
    cov_dat, reads, kinds = synth.pileup_dataset(12, gene_ids, 6, 300, 2500)
    m = GeneNMFOA(degnorm_iter=3, nmf_iter=100, n_jobs=1)
    tr = Tracer(m)
    t1 = time.time()
    est = m.run(cov_dat, reads)
    n = len(gene_ids)
    n_calls, sum_cols, n0 = trace_arrays(tr.calls)
    out.update(run_seed=12, run_p=6, run_l_min=300, run_l_max=2500, run_gene_ids=np.array(gene_ids), run_reads=reads, run_kinds=kinds,
               run_checksum=np.array([checksum(c) for c in cov_dat.values()]), run_rho=m.rho, run_x_adj=m.x_adj,
               run_flags=m.ran_baseline_selection, run_scale_factors=m.scale_factors, run_rho_hist=np.stack(tr.rho_hist),
               run_scale_hist=np.stack(tr.scale_hist), run_n_calls=n_calls.reshape(3, n), run_sum_cols=sum_cols.reshape(3, n),
               run_n0=n0.reshape(3, n), run_est_rowsum=np.vstack([e.sum(axis=1) for e in est]))
    np.savez_compressed(os.path.join(HERE, 'pileup.npz'), **out)
    print('pileup run: %d genes x 3 iterations in %.0f s' % (n, time.time() - t1))



def sec_mpi():
    seed, p, l_min, l_max, n = 7, 4, 200, 1200, 30
    cov_dat, reads, classes = synth.synth_dataset(seed, n, p, l_min, l_max)
    out = dict(seed=seed, p=p, l_min=l_min, l_max=l_max, gene_ids=np.arange(n), reads=reads, degnorm_iter=2,
               nmf_iter=40, checksum=np.array([checksum(c) for c in cov_dat.values()]))
    single = GeneNMFOA(degnorm_iter=2, nmf_iter=40)
    single.run(cov_dat, reads)
    out['single_rho'] = single.rho
    out['single_x_adj'] = single.x_adj
    out['single_flags'] = single.ran_baseline_selection
    for size in (2, 3):
        comm = FakeComm(size)
        results = [None] * size

        def work(r):
            results[r] = ref_mpi.run_gene_nmfoa_mpi(comm.view(r), cov_dat if r == 0 else None, reads,
                                                    degnorm_iter=2, nmf_iter=40)
        # workers need cov_dat only for len(); the reference reads len(cov_dat) on every rank (nmf_mpi.py:598)
        def work_safe(r):
            results[r] = ref_mpi.run_gene_nmfoa_mpi(comm.view(r), cov_dat, reads, degnorm_iter=2, nmf_iter=40)
        ths = [threading.Thread(target=work_safe, args=(r,)) for r in range(size)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        res = results[0]
        out['mpi{0}_rho'.format(size)] = res['rho']
        out['mpi{0}_x_adj'.format(size)] = res['x_adj']
        out['mpi{0}_flags'.format(size)] = res['ran_baseline_selection']
        out['mpi{0}_est_rowsum'.format(size)] = np.vstack([e.sum(axis=1) for e in res['estimates'].values()])
        out['mpi{0}_chunk_lens'.format(size)] = np.array([len(c) for c in split_into_chunks(list(cov_dat.keys()), size)])
    np.savez_compressed(os.path.join(HERE, 'mpi.npz'), **out)
    print('mpi done')


def sec_warm():
    """
    G7: synthetic warm-start directory -> reference load_from_previous (warm_start.py) -> the CLI gene filter
    (__main__.py:219-247, restated here because degnorm.__main__ needs pysam/HTSeq) -> GeneNMFOA.run ->
    save_results; the three CSVs it writes are the fixture.
    """
    import tempfile
    import pandas as pd
    from degnorm.warm_start import load_from_previous
    seed, n_genes, p, l_min, l_max, minimax = 5, 60, 6, 200, 1500, 5
    src = tempfile.mkdtemp(prefix='dn_warm_src_')
    out = tempfile.mkdtemp(prefix='dn_warm_out_')
    _fixtures.write_warm_start_dir(src, seed=seed, n_genes=n_genes, p=p, l_min=l_min, l_max=l_max)
    dat = load_from_previous(src, out)
    gene_cov_dict, read_count_df, genes_df, sample_ids = dat['gene_cov_dict'], dat['read_count_df'], dat['genes_df'], dat['sample_ids']
    loaded_order = list(gene_cov_dict.keys())
    delete_idx = []
    for i in range(genes_df.shape[0]):                                       # __main__.py:223-234
        gene = genes_df.gene.iloc[i]
        cov_mat = gene_cov_dict[gene]
        if (cov_mat.max() < minimax) or (cov_mat.shape[1] <= 1):
            delete_idx.append(i)
            del gene_cov_dict[gene]
    if delete_idx:
        read_count_df = read_count_df.drop(delete_idx, axis=0).reset_index(drop=True)
        genes_df = genes_df.drop(delete_idx, axis=0).reset_index(drop=True)
    m = GeneNMFOA(degnorm_iter=2, nmf_iter=50, downsample_rate=1, n_jobs=1)
    est = m.run(gene_cov_dict, reads_dat=read_count_df[sample_ids].values.astype(float))
    m.save_results(est, gene_manifest_df=genes_df, output_dir=out, sample_ids=sample_ids)
    di = pd.read_csv(os.path.join(out, 'degradation_index_scores.csv'))
    adj = pd.read_csv(os.path.join(out, 'adjusted_read_counts.csv'))
    ran = pd.read_csv(os.path.join(out, 'ran_baseline_selection.csv'))
    import pickle
    with open(os.path.join(out, 'chr1', 'estimated_coverage_matrices_chr1.pkl'), 'rb') as f:
        e1 = pickle.load(f)
    np.savez_compressed(os.path.join(HERE, 'warm.npz'), seed=seed, n_genes=n_genes, p=p, l_min=l_min, l_max=l_max,
                        minimax=minimax, degnorm_iter=2, nmf_iter=50, sample_ids=np.array(sample_ids),
                        loaded_order=np.array(loaded_order), genes=di.gene.values.astype(str), chr=di.chr.values.astype(str),
                        rho=di[sample_ids].values, x_adj=adj[sample_ids].values,
                        ran=ran[['iter_0', 'iter_1']].values.astype(bool),
                        chr1_genes=np.array(list(e1.keys())), chr1_est_rowsum=np.vstack([v.sum(axis=1) for v in e1.values()]))
    print('warm done:', len(loaded_order), 'loaded,', di.shape[0], 'run')


def sec_merge():
    """
    SURVEY 8(f-3): the reference's merge_chrom_coverage (reads_coverage_merge.py:167-372) on a synthetic directory of
    per-sample chromosome CSR vectors.  numpy >= 2 removed the `np.float_` alias the reference uses at :353; it is
    restored here (an alias of float64, nothing else changes) so that the reference function runs unmodified.
    """
    import tempfile
    if not hasattr(np, 'float_'):
        np.float_ = np.float64
    from degnorm.reads_coverage_merge import merge_chrom_coverage
    d = tempfile.mkdtemp(prefix='dn_merge_')
    sample_ids, exon_df = _fixtures.write_chrom_coverage_dir(d)
    out = merge_chrom_coverage(d, sample_ids, exon_df, verbose=False)
    genes = list(out.keys())
    np.savez_compressed(os.path.join(HERE, 'merge.npz'), genes=np.array(genes), sample_ids=np.array(sample_ids),
                        lengths=np.array([out[g].shape[1] for g in genes]),
                        flat=np.concatenate([out[g].reshape(-1) for g in genes]))
    print('merge done:', len(genes), 'genes', sum(out[g].size for g in genes), 'values')


SECTIONS = OrderedDict(kat=sec_kat, genes=sec_genes, run_c1=sec_run_c1, run_c2=sec_run_c2, run_c2_deep=sec_run_c2_deep, mpi=sec_mpi,
                       dsamp=sec_dsamp, warm=sec_warm, merge=sec_merge, sparse=sec_sparse, steps=sec_steps, pileup=sec_pileup)

if __name__ == '__main__':
    import logging
    logging.disable(logging.CRITICAL)
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
