"""
Python face of the CPU parity oracle (oracle/nmfoa_oracle.c) plus a numpy restatement of the outer
DegNorm loop.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (degnorm_amd/) must never import this module.

Parity status: PINNED against golden vectors generated from the real reference in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Reference lines restated here (relative to /root/reference/):
    run() outer loop .................. degnorm/nmf.py:483-601
    par_apply_baseline_selection ...... degnorm/nmf.py:377-406 (rho clip :398-399)
    correct_di_scores ................. degnorm/nmf.py:148-158
    run_gene_nmfoa_mpi (same math) .... degnorm/nmf_mpi.py:555-863
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXIT_NAMES = ('low_cov', 'zero_sample', 'median', 'no_loop', 'refined', 'refine_fallback', 'not_found_fallback')


class _Params(ctypes.Structure):
    _fields_ = [('nmf_iter', ctypes.c_int), ('bins', ctypes.c_int), ('min_high_coverage', ctypes.c_int),
                ('downsample_rate', ctypes.c_int), ('skip_baseline_selection', ctypes.c_int)]


def build(force=False):
    """Compile oracle/liboracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'nmfoa_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'liboracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.dno_trace_len.restype = ctypes.c_int
        _LIB.dno_max_threads.restype = ctypes.c_int
    return _LIB


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def make_params(nmf_iter=100, bins=20, min_high_coverage=50, downsample_rate=1, skip_baseline_selection=False):
    """Apply the constructor rules of GeneNMFOA.__init__ (nmf.py:30-53)."""
    downsample_rate = abs(int(downsample_rate))
    mhc = max(2, abs(int(min_high_coverage)))
    if downsample_rate > 1:
        mhc = 2
    return _Params(abs(int(nmf_iter)), abs(int(bins)), mhc, downsample_rate, int(bool(skip_baseline_selection)))


def rank_one(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    p, n = x.shape
    K = np.zeros(p)
    E = np.zeros(n)
    st = lib().dno_rank_one(_dptr(x), p, n, n, _dptr(K), _dptr(E))
    if st != 0:
        raise ValueError('rank_one status {0}'.format(st))
    return K.reshape(-1, 1), E.reshape(1, -1)


def nmf(x, nmf_iter=100):
    x = np.ascontiguousarray(x, dtype=np.float64)
    p, n = x.shape
    K = np.zeros(p)
    E = np.zeros(n)
    st = lib().dno_nmf(_dptr(x), p, n, int(nmf_iter), _dptr(K), _dptr(E))
    if st != 0:
        raise ValueError('nmf status {0}'.format(st))
    return K.reshape(-1, 1), E.reshape(1, -1)


def ratio_svd(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    p, n = x.shape
    est = np.zeros_like(x)
    st = lib().dno_ratio_svd(_dptr(x), p, n, _dptr(est), None, None)
    if st != 0:
        raise ValueError('ratio_svd status {0}'.format(st))
    return est


def split_into_chunks(length, n):
    start = np.zeros(n + 2, dtype=np.int32)
    nb = lib().dno_split_into_chunks(int(length), int(n), _iptr(start))
    return [list(range(start[b], start[b + 1])) for b in range(nb)]


def baseline_selection(F, ds_start=-1, want_estimate=True, **kw):
    """One gene.  Returns (rho (p,), estimate (p x L) or None, flag bool, trace int array)."""
    F = np.ascontiguousarray(F, dtype=np.float64)
    p, L = F.shape
    prm = make_params(**kw)
    rho = np.zeros(p)
    est = np.zeros_like(F) if want_estimate else None
    flag = ctypes.c_int(0)
    trace = np.zeros(lib().dno_trace_len(), dtype=np.int32)
    lib().dno_baseline_selection(_dptr(F), p, L, ctypes.byref(prm), ctypes.c_long(int(ds_start)), _dptr(rho),
                                 _dptr(est) if want_estimate else None, ctypes.byref(flag), _iptr(trace))
    return rho, est, bool(flag.value), trace


def _gene_ptrs(cov_mats):
    n = len(cov_mats)
    is_f32 = int(cov_mats[0].dtype == np.float32)
    keep = [np.ascontiguousarray(c, dtype=np.float32 if is_f32 else np.float64) for c in cov_mats]
    ptrs = (ctypes.c_void_p * n)(*[c.ctypes.data for c in keep])
    L = np.array([c.shape[1] for c in keep], dtype=np.int64)
    return keep, ptrs, L, is_f32


def baseline_batch(cov_mats, scale, prm, ds_start=None, want_estimates=False, n_threads=0):
    """
    adjust_coverage_curves + baseline_selection over a list of raw coverage matrices.
    Returns (rho n x p *unclipped*, flags bool n, trace n x TRACE_LEN, estimates list or None).
    """
    n = len(cov_mats)
    p = cov_mats[0].shape[0]
    keep, ptrs, L, is_f32 = _gene_ptrs(cov_mats)
    scale = np.ascontiguousarray(scale, dtype=np.float64)
    rho = np.zeros((n, p))
    flags = np.zeros(n, dtype=np.int32)
    tl = lib().dno_trace_len()
    trace = np.zeros((n, tl), dtype=np.int32)
    ests, eptrs = None, None
    if want_estimates:
        ests = [np.zeros((p, int(l))) for l in L]
        eptrs = (ctypes.c_void_p * n)(*[e.ctypes.data for e in ests])
    dsp = None
    if ds_start is not None:
        ds_arr = np.ascontiguousarray(ds_start, dtype=np.int64)
        dsp = ds_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_long))
    lib().dno_baseline_batch(n, p, ptrs, is_f32, L.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), _dptr(scale),
                             ctypes.byref(prm), dsp, _dptr(rho), _iptr(flags), _iptr(trace), eptrs, int(n_threads))
    return rho, flags.astype(bool), trace, ests


def ratio_svd_batch(cov_mats, n_threads=0):
    """Returns (est_sums n x p, cov_sums n x p, status n)."""
    n = len(cov_mats)
    p = cov_mats[0].shape[0]
    keep, ptrs, L, is_f32 = _gene_ptrs(cov_mats)
    est_sums = np.zeros((n, p))
    cov_sums = np.zeros((n, p))
    status = np.zeros(n, dtype=np.int32)
    lib().dno_ratio_svd_batch(n, p, ptrs, is_f32, L.ctypes.data_as(ctypes.POINTER(ctypes.c_long)),
                              _dptr(est_sums), _dptr(cov_sums), _iptr(status), int(n_threads))
    return est_sums, cov_sums, status


def run(cov_mats, reads_dat, degnorm_iter=5, nmf_iter=100, bins=20, min_high_coverage=50, downsample_rate=1,
        skip_baseline_selection=False, ds_starts=None, want_estimates=False, n_threads=0, history=None):
    """
    GeneNMFOA.run (nmf.py:483-601) restated.  cov_mats: list of raw (p x L_g) arrays in gene order.
    ds_starts: optional (degnorm_iter x n) systematic-sample start offsets.
    history: optional dict that receives per-iteration rho / scale_factors / traces.
    Returns dict(rho, x_adj, ran_baseline_selection, scale_factors, norm_factors, x_weighted, estimates, rho_init).
    """
    prm = make_params(nmf_iter, bins, min_high_coverage, downsample_rate, skip_baseline_selection)
    n = len(cov_mats)
    x = np.array(reads_dat, dtype=np.float64)
    ran = np.zeros((n, degnorm_iter), dtype=bool)

    est_sums, cov_sums, _ = ratio_svd_batch(cov_mats, n_threads)                       # nmf.py:522-525
    rho = 1 - (cov_sums / (est_sums + 1))                                               # nmf.py:526
    rho_init = rho.copy()
    low = rho.max(axis=1) < 0.1                                                         # nmf.py:529
    count_sums = x[low, :].sum(axis=0) if np.any(low) else x.sum(axis=0)                # nmf.py:530
    norm = count_sums / np.median(count_sums)                                           # nmf.py:531
    xw = x / norm                                                                       # nmf.py:534
    scale = np.copy(norm)                                                               # nmf.py:535
    if history is not None:
        history['scale_init'] = scale.copy()
        history['rho'] = []
        history['scale'] = []
        history['trace'] = []

    x_adj, ests = None, None
    for i in range(degnorm_iter):
        last = i == degnorm_iter - 1
        rho, flags, trace, e = baseline_batch(cov_mats, scale, prm,
                                              ds_start=None if ds_starts is None else ds_starts[i],
                                              want_estimates=want_estimates and last, n_threads=n_threads)
        if e is not None:
            ests = e
        rho[rho > 0.9] = 0.9                                                            # nmf.py:398
        rho[rho < 0.] = 0.                                                              # nmf.py:399
        ran[:, i] = flags                                                               # nmf.py:403
        x_adj = xw / (1 - rho)                                                          # nmf.py:575
        non_bl = rho.max(axis=1) == 0                                                   # nmf.py:155
        if np.sum(non_bl) > 0:
            rho[non_bl, :] = 1 - (xw.sum(axis=0) / x_adj.sum(axis=0))                   # nmf.py:157-158
        x_adj = xw / (1 - rho)                                                          # nmf.py:581
        norm = x_adj.sum(axis=0) / np.median(x_adj.sum(axis=0))                         # nmf.py:584
        xw = xw / norm                                                                  # nmf.py:587
        scale = scale * norm                                                            # nmf.py:590
        if history is not None:
            history['rho'].append(rho.copy())
            history['scale'].append(scale.copy())
            history['trace'].append(trace.copy())

    return dict(rho=rho, x_adj=x_adj, ran_baseline_selection=ran, scale_factors=scale, norm_factors=norm,
                x_weighted=xw, estimates=ests, rho_init=rho_init)
