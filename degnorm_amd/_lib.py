"""
ctypes binding of libdegnorm_amd.so (include/degnorm_amd.h).  Fails loudly: there is no CPU fallback --
if the HIP library is missing or no gfx950 device is visible, constructing a Device raises.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DN_LIB_PATH: load another build of the SAME library (kernel experiments, tools/variant_ab.sh); never a fallback
LIB_PATH = os.environ.get('DN_LIB_PATH') or os.path.join(_HERE, 'libdegnorm_amd.so')
TRACE_LEN = 48

DN_OK = 0
DN_E_INVALID, DN_E_HIP, DN_E_STATE, DN_E_UNSUPPORTED, DN_E_NO_DEVICE = -1, -2, -3, -4, -5

EXIT_NAMES = ('low_cov', 'zero_sample', 'median', 'no_loop', 'refined', 'refine_fallback', 'not_found_fallback')

_lib = None


class DegnormAmdError(RuntimeError):
    pass


class Params(ctypes.Structure):
    _fields_ = [('nmf_iter', ctypes.c_int32), ('bins', ctypes.c_int32), ('min_high_coverage', ctypes.c_int32),
                ('downsample_rate', ctypes.c_int32), ('skip_baseline_selection', ctypes.c_int32),
                ('want_estimates', ctypes.c_int32), ('reserved', ctypes.c_int32 * 2)]


def load(build_if_missing=False):
    """Load the shared library (optionally building it first).  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build
            build.build_library()
        else:
            raise DegnormAmdError('{0} not found: build it with `python -m degnorm_amd.build` '
                                  '(there is no CPU fallback for the NMF-OA path).'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, i64, dbl = c.c_void_p, c.c_int32, c.c_int64, c.c_double
    P = c.POINTER
    lib.dn_version.restype = c.c_char_p
    lib.dn_last_error.restype = c.c_char_p
    lib.dn_device_count.restype = c.c_int
    lib.dn_p_supported.argtypes = [c.c_int]
    lib.dn_create.argtypes = [c.c_int, P(vp)]
    lib.dn_destroy.argtypes = [vp]
    lib.dn_upload_ragged.argtypes = [vp, i64, i32, P(vp), P(i64), i32, i32, P(i64)]
    lib.dn_upload_packed.argtypes = [vp, i64, i32, P(c.c_float), P(i64)]
    lib.dn_set_downsample_hint.argtypes = [vp, i32]
    lib.dn_set_solver_step_cap.argtypes = [vp, i32]
    lib.dn_set_trace_columns.argtypes = [vp, i32]
    lib.dn_ratio_svd_sums.argtypes = [vp, P(dbl), P(dbl), P(i32)]
    lib.dn_baseline_iteration.argtypes = [vp, P(dbl), P(Params), P(i64), P(dbl), P(i32), P(i32)]
    lib.dn_fetch_estimates.argtypes = [vp, P(dbl)]
    lib.dn_outer_begin.argtypes = [vp, P(dbl), i32]
    lib.dn_init_begin.argtypes = [vp, P(dbl)]
    lib.dn_init_partials.argtypes = [vp, P(dbl)]
    lib.dn_outer_begin_scaled.argtypes = [vp, P(dbl), i32]
    lib.dn_outer_partials.argtypes = [vp, P(dbl)]
    lib.dn_outer_apply.argtypes = [vp, P(dbl), P(dbl), i32]
    lib.dn_fetch_outer.argtypes = [vp, P(dbl), P(dbl), P(dbl), P(c.c_uint8)]
    lib.dn_fetch_rows.argtypes = [vp, i64, P(i64), P(dbl), P(i32)]
    lib.dn_fetch_estimates_subset.argtypes = [vp, i64, P(i64), P(dbl)]
    lib.dn_last_kernel_ms.argtypes = [vp]
    lib.dn_last_kernel_ms.restype = dbl
    lib.dn_main_kernel_name.argtypes = [vp]
    lib.dn_main_kernel_name.restype = c.c_char_p
    lib.dn_last_span_ms.argtypes = [vp]
    lib.dn_last_span_ms.restype = dbl
    lib.dn_last_init_ms.argtypes = [vp]
    lib.dn_last_init_ms.restype = dbl
    lib.dn_init_kernel_name.argtypes = [vp]
    lib.dn_init_kernel_name.restype = c.c_char_p
    lib.dn_split_length.argtypes = [vp]
    lib.dn_split_length.restype = i32
    lib.dn_class_lengths.argtypes = [vp, i32, i32, P(i32), P(i32)]
    lib.dn_tiny_length.argtypes = [vp]
    lib.dn_tiny_length.restype = i32
    lib.dn_class_kernel_ms.argtypes = [vp, c.c_int]
    lib.dn_class_kernel_ms.restype = dbl
    lib.dn_class_kernel_name.argtypes = [vp, c.c_int]
    lib.dn_class_kernel_name.restype = c.c_char_p
    lib.dn_synchronize.argtypes = [vp]
    lib.dn_measure_copy_gbps.argtypes = [vp, i64, c.c_int]
    lib.dn_measure_copy_gbps.restype = dbl
    lib.dn_last_rowmax_ms.argtypes = [vp]
    lib.dn_last_rowmax_ms.restype = dbl
    lib.dn_measure_read_gbps.argtypes = [vp, i64, c.c_int]
    lib.dn_measure_read_gbps.restype = dbl
    lib.dn_assemble_coverage.argtypes = [c.c_int, i64, i32, P(i64), P(vp), P(vp), i64, P(i64), i64, P(i32), P(i64), P(i64), P(i32),
                                         P(c.c_float), P(dbl)]
    lib.dn_assemble_last_error.restype = c.c_char_p
    lib.dn_outer_partials_device.argtypes = [vp, P(vp)]
    lib.dn_comm_unique_id.argtypes = [P(c.c_uint8)]
    lib.dn_comm_create.argtypes = [vp, P(c.c_uint8), i32, i32]
    lib.dn_comm_destroy.argtypes = [vp]
    lib.dn_comm_size.argtypes = [vp]
    lib.dn_comm_size.restype = i32
    lib.dn_comm_library.restype = c.c_char_p
    lib.dn_comm_allreduce.argtypes = [vp, P(dbl), i32]
    lib.dn_init_allreduce.argtypes = [vp, P(dbl)]
    lib.dn_outer_allreduce.argtypes = [vp, P(dbl)]
    _lib = lib
    return lib


def _check(rc):
    if rc == DN_OK:
        return
    msg = load().dn_last_error().decode('utf-8', 'replace')
    if rc == DN_E_INVALID:
        raise ValueError(msg)
    raise DegnormAmdError('degnorm_amd error {0}: {1}'.format(rc, msg))


def _p(arr, ctype):
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


def device_count():
    return int(load().dn_device_count())


class Device:
    """One HIP device holding a resident set of gene coverage matrices."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = ctypes.c_void_p()
        _check(self.lib.dn_create(int(device), ctypes.byref(self.h)))
        self.n = 0
        self.p = 0
        self.lengths = None
        self.inexact = 0
        self.trace_cols = TRACE_LEN

    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.dn_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- upload ------------------------------------------------------------------------------------
    def hint_downsample(self, rate):
        """Announce the take-every rate before an upload (kernel-family choice only; see dn_set_downsample_hint)."""
        _check(self.lib.dn_set_downsample_hint(self.h, int(max(1, rate))))
        return self

    def set_trace_columns(self, cols):
        """Leading trace columns copied back per iteration (8 .. TRACE_LEN; default all).  baseline_iteration then returns n x cols."""
        _check(self.lib.dn_set_trace_columns(self.h, int(cols)))
        self.trace_cols = int(cols)
        return self

    def set_solver_step_cap(self, max_steps):
        """Step cap of one on-chip eigen-solve (default 4000); genes that hit it get per-gene status -4."""
        _check(self.lib.dn_set_solver_step_cap(self.h, int(max_steps)))
        return self

    def upload(self, cov_mats, n_threads=0):
        """cov_mats: list of (p x L_g) float64 or float32 arrays (reference layout, nmf.py:488-490)."""
        n = len(cov_mats)
        if n == 0:
            raise ValueError('no coverage matrices')
        p = int(cov_mats[0].shape[0])
        is_f32 = all(c.dtype == np.float32 for c in cov_mats)
        dt = np.float32 if is_f32 else np.float64
        keep = [np.ascontiguousarray(c, dtype=dt) for c in cov_mats]
        for c in keep:
            if c.ndim != 2:
                raise ValueError('Not all coverage matrices are 2-d arrays!')
            if c.shape[0] != p:
                raise ValueError('coverage matrices disagree on the number of samples')
        ptrs = (ctypes.c_void_p * n)(*[c.ctypes.data for c in keep])
        lengths = np.array([c.shape[1] for c in keep], dtype=np.int64)
        inexact = ctypes.c_int64(0)
        _check(self.lib.dn_upload_ragged(self.h, n, p, ptrs, _p(lengths, ctypes.c_int64), int(is_f32),
                                         int(n_threads), ctypes.byref(inexact)))
        self.n, self.p, self.lengths, self.inexact = n, p, lengths, int(inexact.value)
        return self

    def upload_packed(self, packed, lengths, p):
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        if packed.size != int(lengths.sum()) * int(p):
            raise ValueError('packed size does not match p * sum(lengths)')
        _check(self.lib.dn_upload_packed(self.h, len(lengths), int(p), _p(packed, ctypes.c_float),
                                         _p(lengths, ctypes.c_int64)))
        self.n, self.p, self.lengths, self.inexact = len(lengths), int(p), lengths, 0
        return self

    # -- compute -----------------------------------------------------------------------------------
    def ratio_svd_sums(self, fetch=True):
        """fetch=False leaves the two n x p sums on the device (init_partials reduces them there); returns (None, None, None)."""
        if not fetch:
            _check(self.lib.dn_ratio_svd_sums(self.h, None, None, None))
            return None, None, None
        est = np.zeros((self.n, self.p))
        cov = np.zeros((self.n, self.p))
        status = np.zeros(self.n, dtype=np.int32)
        _check(self.lib.dn_ratio_svd_sums(self.h, _p(est, ctypes.c_double), _p(cov, ctypes.c_double),
                                          _p(status, ctypes.c_int32)))
        return est, cov, status

    def init_status(self):
        """Per-gene status of the initial pass alone (re-runs it; the error path of a device-side initialisation)."""
        status = np.zeros(self.n, dtype=np.int32)
        _check(self.lib.dn_ratio_svd_sums(self.h, None, None, _p(status, ctypes.c_int32)))
        return status

    def init_begin(self, reads):
        x = np.ascontiguousarray(reads, dtype=np.float64)
        if x.shape != (self.n, self.p):
            raise ValueError('reads must be n x p')
        _check(self.lib.dn_init_begin(self.h, _p(x, ctypes.c_double)))

    def init_partials(self):
        out = np.zeros(3 * self.p + 4)
        _check(self.lib.dn_init_partials(self.h, _p(out, ctypes.c_double)))
        return out

    def outer_begin_scaled(self, norm, degnorm_iter):
        nm = np.ascontiguousarray(norm, dtype=np.float64)
        if nm.shape != (self.p,):
            raise ValueError('norm must have one entry per sample')
        self._n_iter = int(degnorm_iter)
        _check(self.lib.dn_outer_begin_scaled(self.h, _p(nm, ctypes.c_double), int(degnorm_iter)))

    def baseline_iteration(self, scale, nmf_iter=100, bins=20, min_high_coverage=50, downsample_rate=1,
                           skip_baseline_selection=False, want_estimates=False, ds_start=None, want_trace=True, fetch=True,
                           trace_out=None):
        """
        Returns (rho n x p unclipped, flags bool n, trace n x TRACE_LEN int32 or None).  fetch=False leaves the DI rows and
        flags on the device (outer_partials / outer_apply / fetch_outer work on them there) and returns (None, None, trace).
        trace_out: an (n x TRACE_LEN) int32 array to fill instead of a fresh one (repeated runs: no new pages to fault in).
        """
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        if scale.shape != (self.p,):
            raise ValueError('scale must have one entry per sample')
        prm = Params(int(nmf_iter), int(bins), int(min_high_coverage), int(downsample_rate),
                     int(bool(skip_baseline_selection)), int(bool(want_estimates)))
        rho = np.zeros((self.n, self.p)) if fetch else None
        flags = np.zeros(self.n, dtype=np.int32) if fetch else None
        trace = None
        if want_trace:
            ok = (trace_out is not None and trace_out.shape == (self.n, self.trace_cols) and trace_out.dtype == np.int32
                  and trace_out.flags['C_CONTIGUOUS'])
            trace = trace_out if ok else np.empty((self.n, self.trace_cols), dtype=np.int32)     # the library writes every entry
        dsp = None
        if ds_start is not None:
            ds_arr = np.ascontiguousarray(ds_start, dtype=np.int64)
            if ds_arr.shape != (self.n,):
                raise ValueError('ds_start must have one entry per gene')
            dsp = _p(ds_arr, ctypes.c_int64)
        _check(self.lib.dn_baseline_iteration(self.h, _p(scale, ctypes.c_double), ctypes.byref(prm), dsp,
                                              _p(rho, ctypes.c_double) if fetch else None, _p(flags, ctypes.c_int32) if fetch else None,
                                              _p(trace, ctypes.c_int32) if want_trace else None))
        return rho, (flags.astype(bool) if fetch else None), trace

    # -- the outer update on the device (dn_outer_*) ---------------------------------------------------
    def outer_begin(self, x_weighted, degnorm_iter):
        xw = np.ascontiguousarray(x_weighted, dtype=np.float64)
        if xw.shape != (self.n, self.p):
            raise ValueError('x_weighted must be n x p')
        self._n_iter = int(degnorm_iter)
        _check(self.lib.dn_outer_begin(self.h, _p(xw, ctypes.c_double), int(degnorm_iter)))

    def outer_partials(self):
        out = np.zeros(3 * self.p + 4)
        _check(self.lib.dn_outer_partials(self.h, _p(out, ctypes.c_double)))
        return out

    def outer_partials_device(self):
        """The partial sums left on the device: (device address, number of doubles), for a collective on the buffer itself."""
        ptr = ctypes.c_void_p()
        _check(self.lib.dn_outer_partials_device(self.h, ctypes.byref(ptr)))
        return int(ptr.value), 3 * self.p + 4

    # -- the collective inside the library (dn_comm_*: RCCL on the handle's stream) ------------------
    COMM_ID_BYTES = 128

    @staticmethod
    def comm_unique_id():
        """Rank 0: the opaque bytes every rank hands to comm_create (ncclGetUniqueId)."""
        buf = np.zeros(Device.COMM_ID_BYTES, dtype=np.uint8)
        _check(load().dn_comm_unique_id(_p(buf, ctypes.c_uint8)))
        return buf

    def comm_create(self, unique_id, rank, size):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        if uid.shape != (Device.COMM_ID_BYTES,):
            raise ValueError('the communicator id is {0} bytes'.format(Device.COMM_ID_BYTES))
        _check(self.lib.dn_comm_create(self.h, _p(uid, ctypes.c_uint8), int(rank), int(size)))

    def comm_destroy(self):
        _check(self.lib.dn_comm_destroy(self.h))

    def comm_size(self):
        return int(self.lib.dn_comm_size(self.h))

    @staticmethod
    def comm_library():
        return load().dn_comm_library().decode()

    def comm_allreduce(self, vec):
        """Sum a small float64 vector over the ranks of this handle's communicator."""
        v = np.array(vec, dtype=np.float64).ravel()
        _check(self.lib.dn_comm_allreduce(self.h, _p(v, ctypes.c_double), int(v.size)))
        return v

    def init_allreduce(self):
        """dn_init_partials of this shard summed over the ranks (3p + 4)."""
        v = np.zeros(3 * self.p + 4)
        _check(self.lib.dn_init_allreduce(self.h, _p(v, ctypes.c_double)))
        return v

    def outer_allreduce(self):
        """dn_outer_partials of this shard summed over the ranks (3p + 4): no host hop before the collective."""
        v = np.zeros(3 * self.p + 4)
        _check(self.lib.dn_outer_allreduce(self.h, _p(v, ctypes.c_double)))
        return v

    def outer_apply(self, avg_di, norm, it):
        norm = np.ascontiguousarray(norm, dtype=np.float64)
        avg = None if avg_di is None else np.ascontiguousarray(avg_di, dtype=np.float64)
        _check(self.lib.dn_outer_apply(self.h, None if avg is None else _p(avg, ctypes.c_double), _p(norm, ctypes.c_double), int(it)))

    def fetch_outer(self, out=None):
        """(rho, x_adj, x_weighted, ran_baseline_selection n x degnorm_iter bool) as the device holds them.
        out: the tuple a previous call returned, to be filled again (repeated runs: no new pages to fault in)."""
        if out is not None and out[0].shape == (self.n, self.p) and out[3].shape == (self.n, self._n_iter):
            rho, x_adj, xw, ran_b = out
            ran = ran_b.view(np.uint8)
        else:
            rho, x_adj, xw = np.empty((self.n, self.p)), np.empty((self.n, self.p)), np.empty((self.n, self.p))
            ran = np.empty((self.n, self._n_iter), dtype=np.uint8)
        _check(self.lib.dn_fetch_outer(self.h, _p(rho, ctypes.c_double), _p(x_adj, ctypes.c_double), _p(xw, ctypes.c_double),
                                       _p(ran, ctypes.c_uint8)))
        return rho, x_adj, xw, ran.view(np.bool_)                       # the device writes 0 / 1

    def fetch_rows(self, rows):
        """Raw (unclipped) DI rows and flags of a few genes of the last baseline_iteration."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        rho = np.empty((rows.size, self.p))
        flags = np.zeros(rows.size, dtype=np.int32)
        _check(self.lib.dn_fetch_rows(self.h, rows.size, _p(rows, ctypes.c_int64), _p(rho, ctypes.c_double), _p(flags, ctypes.c_int32)))
        return rho, flags.astype(bool)

    def fetch_estimates_subset(self, gene_ids):
        """Estimates of the chosen genes only (indices in upload order): list of (p x L_g) float64 arrays."""
        ids = np.ascontiguousarray(gene_ids, dtype=np.int64)
        if ids.ndim != 1 or ids.size == 0:
            raise ValueError('gene_ids must be a non-empty 1-d sequence')
        if ids.min() < 0 or ids.max() >= self.n:
            raise ValueError('gene id out of range')
        uniq, inverse = np.unique(ids, return_inverse=True)             # the C ABI takes each gene once
        flat = np.empty(int(self.lengths[uniq].sum()) * self.p, dtype=np.float64)
        _check(self.lib.dn_fetch_estimates_subset(self.h, uniq.size, _p(uniq, ctypes.c_int64), _p(flat, ctypes.c_double)))
        mats, o = [], 0
        for L in self.lengths[uniq]:
            cnt = self.p * int(L)
            mats.append(flat[o:o + cnt].reshape(self.p, int(L)))
            o += cnt
        return [mats[k] for k in inverse]

    def fetch_estimates_flat(self):
        """All estimates in one float64 buffer: gene g at p * sum(lengths[:g]), p x L_g row-major."""
        total = int(self.lengths.sum()) * self.p
        flat = np.empty(total, dtype=np.float64)
        _check(self.lib.dn_fetch_estimates(self.h, _p(flat, ctypes.c_double)))
        return flat

    def fetch_estimates(self):
        """List of (p x L_g) float64 arrays, views into one flat buffer, in upload order."""
        flat = self.fetch_estimates_flat()
        out, o = [], 0
        for L in self.lengths:
            cnt = self.p * int(L)
            out.append(flat[o:o + cnt].reshape(self.p, int(L)))
            o += cnt
        return out

    # -- measurement -------------------------------------------------------------------------------
    def last_kernel_ms(self):
        return float(self.lib.dn_last_kernel_ms(self.h))

    def last_span_ms(self):
        return float(self.lib.dn_last_span_ms(self.h))

    def last_init_ms(self):
        return float(self.lib.dn_last_init_ms(self.h))

    def init_kernel_name(self):
        return self.lib.dn_init_kernel_name(self.h).decode()

    def main_kernel_name(self):
        return self.lib.dn_main_kernel_name(self.h).decode()

    def split_length(self):
        return int(self.lib.dn_split_length(self.h))

    def class_lengths(self, p, downsample_rate=1):
        """(split_len, tiny_len) of a cohort of p samples on this device, before any upload (0: the class does not exist)."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        _check(self.lib.dn_class_lengths(self.h, int(p), int(downsample_rate), ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def tiny_length(self):
        return int(self.lib.dn_tiny_length(self.h))

    def class_kernel_ms(self, cls):
        return float(self.lib.dn_class_kernel_ms(self.h, int(cls)))

    def class_kernel_name(self, cls):
        return self.lib.dn_class_kernel_name(self.h, int(cls)).decode()

    def measure_copy_gbps(self, nbytes=1 << 30, reps=5):
        return float(self.lib.dn_measure_copy_gbps(self.h, int(nbytes), int(reps)))

    def measure_read_gbps(self, nbytes=1 << 30, reps=5):
        """Stream-read ceiling of the device (GB/s)."""
        return float(self.lib.dn_measure_read_gbps(self.h, int(nbytes), int(reps)))

    def last_rowmax_ms(self):
        return float(self.lib.dn_last_rowmax_ms(self.h))
