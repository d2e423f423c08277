"""
Randomised differential run of the device path against the CPU oracle (test infrastructure, like tests/): seeded batches of
genes with random sample counts, lengths, depth regimes and pathological structure (empty samples, empty stretches,
piecewise-constant coverage = exact ties, counts beyond 16 bits, non-integer coverage, strong 3' decay), random scale
factors, nmf_iter, bins, min_high_coverage and down-sampling.  Every batch is compared like the parity tests: branch trace
and flags exact, DI and estimates to 1e-8.  Mismatching genes are listed with their traces (a tie flip shows as a trace
that differs by one column / one drop; anything else is a bug to chase).
usage (GPU box): python tools/fuzz_parity.py [--rounds 40] [--seed 1] [--out gpurun_out/fuzz/fuzz.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_gene(rng, p, L, kind):
    env = 1.0 + np.abs(np.sin(np.linspace(0, rng.uniform(0.5, 7), L) + rng.uniform(0, 3)))
    depth = float(rng.choice([0.3, 3.0, 30.0, 300.0, 3000.0]))
    mean = depth * np.outer(rng.lognormal(0, 0.5, p), env)
    if kind == 'plain':
        return rng.poisson(mean).astype(float)
    if kind == 'decay':                                    # 3' bias in some samples: the drop-bin loop runs long
        for i in range(p):
            if rng.random() < 0.5:
                mean[i] *= np.linspace(rng.uniform(0.02, 0.8), 1.0, L) ** rng.uniform(0.5, 3)
        return rng.poisson(mean).astype(float)
    if kind == 'empty_sample':
        x = rng.poisson(mean).astype(float)
        x[rng.integers(0, p)] = 0.0
        return x
    if kind == 'holes':                                    # stretches without coverage in every sample
        x = rng.poisson(mean).astype(float)
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(1, max(2, L // 3))))
            x[:, a:b] = 0.0
        return x
    if kind == 'steps':                                    # piecewise constant, small integers: exact ties everywhere
        k = int(rng.integers(1, 9))
        edges = np.sort(rng.integers(0, L, size=k))
        lvl = rng.integers(0, 6, size=(p, k + 1)).astype(float)
        x = np.zeros((p, L))
        prev = 0
        for j, e in enumerate(list(edges) + [L]):
            x[:, prev:e] = lvl[:, j:j + 1]
            prev = e
        return x
    if kind == 'big':                                      # counts beyond 16 bits: the variant without packed counts
        x = rng.poisson(mean).astype(float)
        x[rng.integers(0, p), rng.integers(0, L, size=max(1, L // 50))] = float(rng.integers(65536, 400000))
        return x
    if kind == 'fractional':                               # not whole numbers (still exact in fp32)
        return rng.poisson(mean).astype(float) + rng.choice([0.0, 0.25, 0.5], size=(p, L))
    if kind == 'rank1':
        return np.outer(rng.integers(1, 9, size=p), rng.integers(0, 50, size=L)).astype(float)
    if kind == 'spike':                                    # one very deep base
        x = rng.poisson(mean).astype(float)
        x[:, rng.integers(0, L)] *= 40.0
        return x
    raise ValueError(kind)


KINDS = ['plain', 'plain', 'decay', 'decay', 'empty_sample', 'holes', 'steps', 'big', 'fractional', 'rank1', 'spike']


def tier_filling_genes(rng, p):
    """Flat, deep coverage (every base an active column) of exactly the register tier's capacity, one column less and one more -- of a
    wavefront (pair class: where the straight-line body of csrc/dn_kernels.hpp starts) and of a 128-thread workgroup (narrow class)."""
    rt_cols = min(12, 256 // (2 * p + (p + 1) // 2))                   # dn_kernels.hpp rt_cols<P, X16 = true>
    out = []
    for cap in (rt_cols * 64, rt_cols * 128):
        for L in (cap - 1, cap, cap + 1):
            out.append(rng.poisson(np.outer(rng.uniform(150., 400., p), np.ones(L))).astype(float))
    return out


def one_round(rng, device_cls, oracle, log, budget_cols, kinds_menu=None, force_p=None, tier_fill=False):
    p = int(force_p) if force_p else int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 11, 12, 13, 14, 15, 16, 17, 19, 24, 32, 33, 47, 50, 64]))
    rate = 1 if tier_fill else int(rng.choice([1, 1, 1, 1, 40, 200, 500]))
    n_genes = int(rng.integers(1, 90))
    covs, kinds = [], []
    if tier_fill:
        covs = tier_filling_genes(rng, p)
        kinds = ['tier_fill'] * len(covs)
        n_genes += len(covs)
    for g in range(len(covs), n_genes):
        if rate > 1:
            L = int(rng.integers(rate + 1, min(13 * rate, 6000) + 1)) if rng.random() < 0.8 else int(rng.integers(rate + 1, 6001))   # the reference refuses rate >= a gene's length
        else:
            r = rng.random()
            L = int(rng.integers(1, 130)) if r < 0.15 else int(rng.integers(130, 2500)) if r < 0.8 else int(rng.integers(2500, budget_cols))
        kind = str(rng.choice(kinds_menu or KINDS))
        covs.append(make_gene(rng, p, L, kind))
        kinds.append(kind)
    scale = np.exp(rng.uniform(-1.2, 1.2, p)) if rng.random() < 0.5 else np.linspace(0.9, 1.15, p)
    T = int(rng.choice([1, 2, 5, 9, 20, 40]))
    bins = int(rng.choice([2, 5, 20, 20, 20, 33]))
    mhc = int(rng.choice([2, 10, 50, 50])) if rate == 1 else 2
    skip = bool(rng.random() < 0.1)
    offs = rng.integers(0, rate, size=n_genes).astype(np.int64) if rate > 1 else None
    what = 'p={0} genes={1} rate={2} T={3} bins={4} mhc={5} skip={6} L=[{7},{8}]'.format(
        p, n_genes, rate, T, bins, mhc, skip, min(c.shape[1] for c in covs), max(c.shape[1] for c in covs))
    dev = device_cls(0)
    try:
        if rate > 1 and rng.random() < 0.7:
            dev.hint_downsample(rate)
        dev.upload(covs)
        init = dev.ratio_svd_sums() if hasattr(dev, 'ratio_svd_sums') else None      # the initial DI pass (nmf.py:109-121) on the same upload
        kw = dict(nmf_iter=T, bins=bins, min_high_coverage=mhc, skip_baseline_selection=skip, want_estimates=True)
        if rate > 1:
            kw.update(downsample_rate=rate, ds_start=offs)
        rho, flags, trace = dev.baseline_iteration(scale, **kw)
        est = dev.fetch_estimates()
        names = [dev.class_kernel_name(k) for k in range(3)]
    finally:
        dev.close()
    prm = oracle.make_params(nmf_iter=T, bins=bins, min_high_coverage=mhc, skip_baseline_selection=skip, downsample_rate=rate)
    okw = dict(want_estimates=True)
    if rate > 1:
        okw['ds_start'] = offs
    rho_o, flags_o, trace_o, est_o = oracle.baseline_batch(covs, scale, prm, **okw)
    bad = []
    cols = [0, 1, 2, 3, 5, 6]
    init_o = oracle.ratio_svd_batch(covs) if init is not None else None
    for g in range(n_genes):
        why = []
        if init is not None:
            (e_d, c_d, s_d), (e_o, c_o, s_o) = init, init_o
            if s_d[g] != s_o[g]:
                why.append('initial pass status dev {0} oracle {1}'.format(s_d[g], s_o[g]))
            elif s_o[g] == 0 and not (np.allclose(c_d[g], c_o[g], rtol=1e-13) and np.allclose(e_d[g], e_o[g], rtol=1e-9, atol=1e-7 * max(1.0, float(np.max(e_o[g]))))):
                why.append('initial pass sums: max rel diff {0:.2e}'.format(float(np.max(np.abs(e_d[g] - e_o[g]) / np.maximum(np.abs(e_o[g]), 1e-30)))))
        if not np.array_equal(trace[g, cols], trace_o[g, cols]):
            why.append('trace dev {0} oracle {1}'.format(trace[g, :7].tolist(), trace_o[g, :7].tolist()))
        if flags[g] != flags_o[g]:
            why.append('flag dev {0} oracle {1}'.format(flags[g], flags_o[g]))
        if not why:
            if not np.allclose(rho[g], rho_o[g], rtol=1e-8, atol=1e-10, equal_nan=True):
                why.append('rho max abs diff {0:.3e}'.format(np.nanmax(np.abs(rho[g] - rho_o[g]))))
            if est_o is not None and est_o[g] is not None and not np.allclose(est[g], est_o[g], rtol=1e-8, atol=1e-8, equal_nan=True):
                why.append('estimate max abs diff {0:.3e}'.format(np.nanmax(np.abs(est[g] - est_o[g]))))
        if why:
            bad.append((g, kinds[g], covs[g].shape[1], why))
    log('{0}  kernels {1}  -> {2} mismatching genes'.format(what, [n for n in names if n], len(bad)))
    for g, kind, L, why in bad:
        log('      gene {0} ({1}, L={2}): {3}'.format(g, kind, L, '; '.join(why)))
    return n_genes, len(bad)


def one_chain_round(rng, oracle, log, budget_cols):
    """The whole chain (GeneNMFOA.fit: initial pass, normalisation, outer iterations with the device-side update) against oracle.run."""
    from collections import OrderedDict
    from degnorm_amd.nmf import GeneNMFOA
    from degnorm_amd import synth
    p = int(rng.choice([2, 3, 4, 6, 8, 10, 10, 12, 13, 16, 17, 24, 33, 50, 64]))
    rate = int(rng.choice([1, 1, 1, 40, 200]))
    n_genes = int(rng.integers(2, 120))
    iters = int(rng.choice([1, 2, 3, 5]))
    T = int(rng.choice([2, 5, 12, 30]))
    covs, kinds = [], []
    for g in range(n_genes):
        if rate > 1:
            L = int(rng.integers(rate + 1, min(13 * rate, 6000) + 1))
        else:
            r = rng.random()
            L = int(rng.integers(60, 400)) if r < 0.3 else int(rng.integers(400, 2500)) if r < 0.85 else int(rng.integers(2500, budget_cols))
        kind = str(rng.choice(['plain', 'plain', 'decay', 'decay', 'holes', 'big', 'fractional', 'spike', 'empty_sample']))
        covs.append(make_gene(rng, p, L, kind))
        kinds.append(kind)
    reads = np.vstack([synth.read_counts_from_coverage(c) for c in covs])
    ds = rng.integers(0, rate, size=(iters, n_genes)).astype(np.int64) if rate > 1 else None
    what = 'chain p={0} genes={1} rate={2} iters={3} T={4}'.format(p, n_genes, rate, iters, T)
    kw = dict(degnorm_iter=iters, nmf_iter=T, downsample_rate=rate, min_high_coverage=2 if rate > 1 else 50)
    hist = {}
    try:
        ref = oracle.run(covs, reads, ds_starts=ds, n_threads=8, history=hist, **kw)
        ref_err = None
    except Exception as e:                                   # the reference's own errors (e.g. a sample without reads)
        ref, ref_err = None, type(e).__name__ + ': ' + str(e)[:80]
    m = GeneNMFOA(degnorm_iter=iters, nmf_iter=T, downsample_rate=rate, device=0)
    if ds is not None:
        m.downsample_offsets = ds
    try:
        m.fit(OrderedDict(('g%06d' % k, c) for k, c in enumerate(covs)), reads)
        dev_err = None
    except Exception as e:
        dev_err = type(e).__name__ + ': ' + str(e)[:80]
    if ref_err or dev_err:
        same = (ref_err is not None) and (dev_err is not None) and ref_err.split(':')[0] == dev_err.split(':')[0]
        log('{0}  -> oracle {1} / device {2}  [{3}]'.format(what, ref_err, dev_err, 'same error class' if same else 'DIFFERENT'))
        return n_genes, 0 if same else n_genes
    flipped = np.zeros(n_genes, dtype=bool)
    for i in range(iters):
        flipped |= np.any(m.traces[i][:, [0, 1, 2, 3, 5, 6]] != hist['trace'][i][:, [0, 1, 2, 3, 5, 6]], axis=1)
    flipped |= np.any(m.ran_baseline_selection != ref['ran_baseline_selection'], axis=1)
    ok = ~flipped
    rel = np.abs(m.rho - ref['rho']) / np.maximum(np.abs(ref['rho']), 1e-6)
    rel_adj = np.abs(m.x_adj - ref['x_adj']) / np.maximum(np.abs(ref['x_adj']), 1e-300)
    sf = float(np.max(np.abs(m.scale_factors - ref['scale_factors']) / ref['scale_factors']))
    worst = float(rel[ok].max()) if ok.any() else 0.0
    worst_adj = float(rel_adj[ok].max()) if ok.any() else 0.0
    bad = int(flipped.sum()) + int(((rel > 1e-5).any(axis=1) & ok).sum())
    log('{0}  -> flipped {1}, unflipped max rel DI {2:.1e}, adjusted counts {3:.1e}, scale factors {4:.1e}'.format(what, int(flipped.sum()), worst, worst_adj, sf))
    for k in np.flatnonzero(flipped)[:3]:
        for i in range(iters):
            if np.any(m.traces[i][k, :7] != hist['trace'][i][k, :7]):
                log('      gene {0} ({1}, L={2}) iteration {3}: trace dev {4} oracle {5}'.format(k, kinds[k], covs[k].shape[1], i + 1, m.traces[i][k, :7].tolist(), hist['trace'][i][k, :7].tolist()))
                break
    return n_genes, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=40)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--max-length', type=int, default=9000)
    ap.add_argument('--out', default='')
    ap.add_argument('--chain', action='store_true', help='whole-chain rounds (GeneNMFOA.fit vs oracle.run) instead of single baseline iterations')
    args = ap.parse_args()
    from degnorm_amd import _lib
    from oracle import oracle
    oracle.build()
    fh = open(args.out, 'w') if args.out else None

    def log(s):
        print(s, flush=True)
        if fh:
            fh.write(s + '\n'); fh.flush()

    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    tot = bad = 0
    for r in range(args.rounds):
        lg = lambda s, r=r: log('[{0:3d}] '.format(r) + s if not s.startswith('      ') else s)
        n, b = one_chain_round(rng, oracle, lg, args.max_length) if args.chain else one_round(rng, _lib.Device, oracle, lg, args.max_length)
        tot += n; bad += b
    log('fuzz: seed {0}, {1} rounds, {2} genes, {3} mismatching, {4:.0f} s'.format(args.seed, args.rounds, tot, bad, time.time() - t0))
    return 0


if __name__ == '__main__':
    sys.exit(main())
