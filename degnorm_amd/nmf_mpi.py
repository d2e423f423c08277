"""
Gene-sharded outer DegNorm iteration -- the MI355X counterpart of degnorm/nmf_mpi.py.

The reference partitions genes over MPI ranks (nmf_mpi.py:603-629), has rank 0 re-scale and re-send every
coverage matrix each iteration (:745-760) and collect every estimate and DI row (:796-815): ~GBs of pickle
through rank 0 per iteration although the only cross-gene coupling is a length-p vector.  Here each rank
(one process per GPU) uploads its contiguous gene chunk once, keeps it resident in HBM, and per outer
iteration contributes 3p + 4 float64 (three per-sample partial sums and four counters) to ONE all-reduce (RCCL over
xGMI when the process group is "nccl"; gloo on CPU for tests):

    A[i] = sum over genes with rho.max() > 0 of x_w[g,i] / (1 - rho[g,i])
    B[i] = sum over genes with rho.max() == 0 of x_w[g,i]            (genes correct_di_scores rewrites)
    W[i] = sum over all genes of x_w[g,i]

from which every rank computes identically (nmf.py:148-158, :575-590; nmf_mpi.py:821-838):

    S_pre = A + B;  avg_di = 1 - W / S_pre;  S_post = A + B * S_pre / W;
    norm = S_post / median(S_post);  x_w /= norm;  scale *= norm

`run_gene_nmfoa_mpi(comm, cov_dat, reads_dat, ...)` keeps the reference's signature and return value
(nmf_mpi.py:555-580, :852-863); `comm` may be a TorchComm (below) or any object with
.size/.rank/.send/.recv/.Barrier (mpi4py duck type, as in the reference), optionally .allreduce.
"""
import logging
import os
from collections import OrderedDict

import numpy as np

from . import _lib
from .utils import split_into_chunks, partition_by_cost, measured_gene_cost, rebalance_moves
from .results import write_results

__all__ = ['run_gene_nmfoa_mpi', 'save_results', 'ShardedNMFOA', 'TorchComm', 'LocalComm']


# ----------------------------------------------------------------------------------------------- #
# communicators
# ----------------------------------------------------------------------------------------------- #
class LocalComm(object):
    """Single-process communicator (size 1)."""
    size = 1
    rank = 0

    def allreduce_sum(self, vec):
        return np.array(vec, dtype=np.float64)

    def Barrier(self):
        pass

    def gather_objects(self, obj):
        return [obj]


class TorchComm(object):
    """
    torch.distributed process group as a communicator: backend "nccl" is RCCL on ROCm (one process per
    GPU, small float64 tensors on the device), "gloo" runs the same code path on CPU for tests.
    Exposes the mpi4py-style surface the reference uses (.size .rank .send .recv .Barrier) plus
    allreduce_sum / gather_objects.
    """

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        if device is None:
            device = 'cuda:{0}'.format(int(os.environ.get('LOCAL_RANK', 0))) if self.backend == 'nccl' else 'cpu'
        self.device = torch.device(device)

    def allreduce_sum(self, vec):
        t = self.torch.as_tensor(np.ascontiguousarray(vec, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def allreduce_device(self, ptr, n, device_index):
        """
        Sum n float64 that already sit in device memory at `ptr` over all ranks, in place (RCCL reads and writes the library's
        own buffer over xGMI), and return the totals as a numpy vector.  None if this communicator cannot (gloo, other device).
        """
        if (self.backend != 'nccl' or self.device.type != 'cuda' or (self.device.index or 0) != int(device_index)
                or getattr(self, '_no_device_view', False)):
            return None

        class _Buf(object):                                           # zero-copy view of foreign device memory
            __cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}
        try:
            t = self.torch.as_tensor(_Buf(), device=self.device)
        except Exception as e:                                        # this torch cannot adopt the pointer: host staging from now on
            logging.info('device-side all-reduce unavailable ({0}); staging through the host'.format(e))
            self._no_device_view = True
            return None
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.device_reductions = getattr(self, 'device_reductions', 0) + 1
        return t.cpu().numpy()

    def Barrier(self):
        if self.backend == 'nccl':
            self.dist.barrier(group=self.group, device_ids=[self.device.index or 0])
        else:
            self.dist.barrier(group=self.group)

    def gather_objects(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def bcast_object(self, obj, root=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=root, group=self.group)
        return box[0]

    # raw buffers, point to point: coverage shares and results travel as bytes (chunked dist.send / dist.recv on uint8 views --
    # gloo: straight from / into the numpy memory; nccl: staged through the device, RCCL p2p over xGMI), never as pickles
    CHUNK_BYTES = 1 << 28
    bytes_sent = 0
    bytes_received = 0

    def send_array(self, arr, dest, pending=None):
        """`pending` (a list): do not wait -- the (request, buffer) pairs are appended to it and the caller waits with wait_sends()
        (rank 0 packs the next rank's share while this one travels)."""
        a = np.ascontiguousarray(arr)
        flat = a.reshape(-1).view(np.uint8)
        for lo in range(0, flat.size, self.CHUNK_BYTES):
            piece = flat[lo:lo + self.CHUNK_BYTES]
            if not piece.flags.writeable:                             # a memory-mapped side-car: torch wants a writable buffer
                piece = np.array(piece)
            t = self.torch.from_numpy(piece)
            if self.backend == 'nccl':
                t = t.to(self.device)
            if pending is None:
                self.dist.send(t, dst=dest, group=self.group)
            else:
                pending.append((self.dist.isend(t, dst=dest, group=self.group), t, a))
        self.bytes_sent += int(flat.size)

    @staticmethod
    def wait_sends(pending):
        for req, _, _ in pending:
            req.wait()
        del pending[:]

    def recv_array(self, shape, dtype, source):
        out = np.empty(shape, dtype=dtype)
        flat = out.reshape(-1).view(np.uint8)
        for lo in range(0, flat.size, self.CHUNK_BYTES):
            piece = flat[lo:lo + self.CHUNK_BYTES]
            if self.backend == 'nccl':
                t = self.torch.empty(piece.size, dtype=self.torch.uint8, device=self.device)
                self.dist.recv(t, src=source, group=self.group)
                piece[...] = t.cpu().numpy()
            else:
                self.dist.recv(self.torch.from_numpy(piece), src=source, group=self.group)
        self.bytes_received += int(flat.size)
        return out

    # point-to-point object passing, for callers that drive this like an mpi4py communicator
    def send(self, obj, dest, tag=0):
        self.dist.send_object_list([obj], dst=dest, group=self.group)

    def recv(self, source, tag=0):
        box = [None]
        self.dist.recv_object_list(box, src=source, group=self.group)
        return box[0]


def _allreduce(comm, vec):
    """Sum a small float64 vector over all ranks of `comm` (native collective when it has one)."""
    if hasattr(comm, 'allreduce_sum'):
        return comm.allreduce_sum(vec)
    vec = np.ascontiguousarray(vec, dtype=np.float64)
    if comm.size == 1:
        return vec
    if hasattr(comm, 'allreduce'):            # mpi4py
        return np.asarray(comm.allreduce(vec))
    # bare send/recv communicator (the reference's duck type): fold through rank 0
    if comm.rank == 0:
        total = vec.copy()
        for r in range(1, comm.size):
            total += comm.recv(source=r, tag=777 + r)
        for r in range(1, comm.size):
            comm.send(total, dest=r, tag=888 + r)
        return total
    comm.send(vec, dest=0, tag=777 + comm.rank)
    return comm.recv(source=0, tag=888 + comm.rank)


def _bcast(comm, obj):
    """Rank 0's `obj` on every rank."""
    if comm.size == 1:
        return obj
    if hasattr(comm, 'bcast_object'):
        return comm.bcast_object(obj, root=0)
    if hasattr(comm, 'bcast'):                # mpi4py
        return comm.bcast(obj, root=0)
    if comm.rank == 0:
        for r in range(1, comm.size):
            comm.send(obj, dest=r, tag=555 + r)
        return obj
    return comm.recv(source=0, tag=555 + comm.rank)


def _send_arrays(comm, dest, tag, arrays, pending=None):
    """numpy arrays to one rank: as raw buffers when the communicator can (TorchComm.send_array; with `pending` without waiting),
    else as one object (the reference's own way: mpi4py pickles, nmf_mpi.py:627)."""
    if hasattr(comm, 'send_array'):
        for a in arrays:
            if pending is not None and hasattr(comm, 'wait_sends'):
                comm.send_array(a, dest, pending)
            else:
                comm.send_array(a, dest)
    else:
        comm.send(tuple(np.ascontiguousarray(a) for a in arrays), dest=dest, tag=tag)


def _recv_arrays(comm, source, tag, specs):
    """The receiving side of _send_arrays: specs = [(shape, dtype), ...] (the receiver knows them from the partition)."""
    if hasattr(comm, 'recv_array'):
        return [comm.recv_array(shape, dtype, source) for shape, dtype in specs]
    got = comm.recv(source=source, tag=tag)
    return [np.asarray(a, dtype=dt).reshape(shape) for a, (shape, dt) in zip(got, specs)]


def _gather(comm, obj):
    if hasattr(comm, 'gather_objects'):
        return comm.gather_objects(obj)
    if comm.size == 1:
        return [obj]
    if comm.rank == 0:
        return [obj] + [comm.recv(source=r, tag=999 + r) for r in range(1, comm.size)]
    comm.send(obj, dest=0, tag=999 + comm.rank)
    return None


# ----------------------------------------------------------------------------------------------- #
class ShardedNMFOA(object):
    """
    One rank's share of a DegNorm run: a resident gene shard on one GPU plus the replicated per-sample state.
    A rank may hold NO genes (more ranks than gene chunks, nmf_mpi.py:613): it then contributes zeros to every collective.
    """

    def __init__(self, comm=None, device=None, degnorm_iter=5, downsample_rate=1, min_high_coverage=50,
                 nmf_iter=100, bins=20, skip_baseline_selection=False, random_state=123):
        self.comm = comm if comm is not None else LocalComm()
        self.degnorm_iter = abs(int(degnorm_iter))
        self.nmf_iter = abs(int(nmf_iter))
        self.bins = abs(int(bins))
        self.min_high_coverage = max(2, abs(int(min_high_coverage)))
        self.downsample_rate = abs(int(downsample_rate))
        if self.downsample_rate > 1:
            self.min_high_coverage = 2                    # nmf_mpi.py:594-596
        self.skip_baseline_selection = skip_baseline_selection
        self.random_state = random_state
        self._dev_id = int(os.environ.get('LOCAL_RANK', 0)) if device is None else int(device)
        self._dev = None           # the HIP device, opened on first use; no fallback: without the library or a GPU it raises
        self.kernel_ms = []
        self.traces = []
        self.class_ms = []
        self.span_ms = []
        self.scale_hist, self.rho_raw_hist, self.flags_hist, self.n_failed = [], [], [], []
        self.offsets_hist = []
        self.history_rows = None                          # optional: local gene rows whose raw DI / flags are kept per iteration
        self.device_outer = True                          # False: the outer update in numpy on the host (same results)
        self._device_outer = False
        self._state_on_device = False
        self.downsample_offsets = None                    # optional (degnorm_iter x n_local) explicit starts
        self.n_local = 0
        self.n_total = 0
        self.global_ids = None
        self.p = 0
        self.trace_columns = None                         # leading trace columns fetched per iteration (None: all; >= 8: the counters)
        self.reuse_buffers = False                        # True: repeated runs fill the same host arrays (traces, final state) again
        self._trace_bufs, self._state_bufs = {}, None
        self.gene_names = None                            # optional: names of the local genes, for error / warning texts
        self.n_flagged = []                               # per outer iteration: genes sent through baseline selection (all ranks)
        self._lib_comm = False                            # True: the collectives run inside the library (attach_library_comm)
        self.library_reductions = 0
        self.keep_packed = False                          # True: the packed float32 shard stays on the host too (redeal() needs it)
        self.redeal_after_first = False                   # True: run() levels the measured per-gene cost over the ranks after iteration 1
        self._packed, self._lengths = None, None
        self.redeal_info = None
        self._ran_saved = None                            # flags columns of the iterations before a re-deal (the device restarts its own)

    @property
    def dev(self):
        if self._dev is None:
            self._dev = _lib.Device(self._dev_id)
        return self._dev

    # -- the collective inside the library ---------------------------------------------------------
    def attach_library_comm(self):
        """
        From here on every collective of this engine runs INSIDE the library (include/degnorm_amd.h dn_comm_*: RCCL on the library's
        own stream, in place on its device buffer -- no tensor, no host hop before the reduction): rank 0 draws the communicator id,
        `self.comm` only carries those 128 bytes to the other ranks once.  Collective: every rank of `self.comm` must call it.
        """
        uid = _lib.Device.comm_unique_id() if self.comm.rank == 0 else None
        uid = _bcast(self.comm, None if uid is None else uid.tobytes())
        self.dev.comm_create(np.frombuffer(uid, dtype=np.uint8), self.comm.rank, self.comm.size)
        self._lib_comm = True

    def _sum(self, vec):
        """Sum a small float64 vector over the ranks."""
        if self._lib_comm:
            self.library_reductions += 1
            return self.dev.comm_allreduce(vec)
        return _allreduce(self.comm, vec)

    # -- data -------------------------------------------------------------------------------------
    def load(self, cov_mats, reads, global_ids=None, n_total=None, p=None, n_threads=0):
        """cov_mats: this rank's (p x L_g) matrices; global_ids: their positions in the whole data set (for the
        partition-invariant down-sampling offsets); n_total: genes over all ranks; n_threads: host packing threads
        (0: as many as the box has, at most 16)."""
        if len(cov_mats) > 0:
            if self.trace_columns is not None and hasattr(self.dev, 'set_trace_columns'):
                self.dev.set_trace_columns(self.trace_columns)
            self.dev.hint_downsample(self.downsample_rate)
            self.dev.upload(cov_mats, n_threads=n_threads)
            if getattr(self.dev, 'inexact', 0):
                logging.warning('{0} coverage values are not exactly representable in float32; they were rounded '
                                'on upload.'.format(self.dev.inexact))
            p = self.dev.p
        self._set_reads(reads, len(cov_mats), p, global_ids, n_total)

    def load_packed(self, packed, lengths, p, reads, global_ids=None, n_total=None):
        if self.keep_packed:
            self._packed, self._lengths = np.ascontiguousarray(packed, dtype=np.float32), np.asarray(lengths, dtype=np.int64).copy()
        if len(lengths) > 0:
            if self.trace_columns is not None and hasattr(self.dev, 'set_trace_columns'):
                self.dev.set_trace_columns(self.trace_columns)
            self.dev.hint_downsample(self.downsample_rate)
            self.dev.upload_packed(packed, lengths, p)
        self._set_reads(reads, len(lengths), p, global_ids, n_total)

    def _set_reads(self, reads, n_local, p, global_ids, n_total):
        if p is None:
            raise ValueError('an empty shard needs the sample count p')
        self.n_local, self.p = int(n_local), int(p)
        self.x = np.array(reads, dtype=np.float64).reshape(self.n_local, self.p)
        self.global_ids = np.arange(self.n_local) if global_ids is None else np.asarray(global_ids, dtype=np.int64)
        if self.global_ids.shape != (self.n_local,):
            raise ValueError('global_ids must name every local gene')
        self.n_total = int(n_total) if n_total is not None else self.n_local
        self._reads_on_device = False
        if self.n_local > 0 and self.device_outer and hasattr(self.dev, 'init_begin'):
            self.dev.init_begin(self.x)              # the read counts live next to the coverage: the initial normalisation runs there
            self._reads_on_device = True

    # -- algorithm --------------------------------------------------------------------------------
    def initialize(self):
        """ratio-SVD DI scores and the initial normalisation factors (nmf.py:521-535, nmf_mpi.py:681-718)."""
        p = self.p
        # With the device-side update the two n x p sums of the initial pass never leave the GPU either: it reduces rho0, the
        # low genes and their read counts itself (dn_init_partials) and the host sees 2p + 2 numbers.
        on_device = self.device_outer and self.n_local > 0 and getattr(self, '_reads_on_device', False) and hasattr(self.dev, 'outer_begin')
        tot = None
        if on_device:
            self.dev.ratio_svd_sums(fetch=False)
            if self._lib_comm:                                         # ONE collective, on the sums where they are (dn_init_allreduce)
                tv = self.dev.init_allreduce()
                self.library_reductions += 1
                n_bad, tot = float(tv[3 * p + 1]), np.concatenate([tv[:p], tv[p:2 * p], [tv[3 * p]]])
                n_bad_local = float(self.dev.init_partials()[3 * p + 1]) if n_bad > 0 else 0.0
            else:
                pv = self.dev.init_partials()
                n_bad_local, part = float(pv[3 * p + 1]), np.concatenate([pv[:p], pv[p:2 * p], [pv[3 * p]]])
            self.rho = None                                            # rho0 stays on the device
        else:
            if self.n_local > 0:
                est_sums, cov_sums, status = self.dev.ratio_svd_sums()
            else:
                est_sums, cov_sums, status = np.zeros((0, p)), np.zeros((0, p)), np.zeros(0, dtype=np.int32)
            n_bad_local = float(np.sum(status != 0))
            self.rho = 1 - (cov_sums / (est_sums + 1))
            low = self.rho.max(axis=1) < 0.1 if self.n_local > 0 else np.zeros(0, dtype=bool)
            part = np.concatenate([self.x[low].sum(axis=0), self.x.sum(axis=0), [float(low.sum())]])
            if self._lib_comm:                                         # the same 3p + 4 layout the device-side ranks reduce
                tv = self._sum(np.concatenate([part[:2 * p], np.zeros(p), [part[2 * p], n_bad_local, 0., 0.]]))
                n_bad, tot = float(tv[3 * p + 1]), np.concatenate([tv[:p], tv[p:2 * p], [tv[3 * p]]])
        if tot is None:
            n_bad = _allreduce(self.comm, [n_bad_local])[0]
        if n_bad > 0:                                                 # every rank raises together
            raise ValueError(self._init_failure_text(int(n_bad), n_bad_local, None if on_device else status))
        if tot is None:
            tot = _allreduce(self.comm, part)
        count_sums = tot[:p] if tot[2 * p] > 0 else tot[p:2 * p]
        self.norm_factors = count_sums / np.median(count_sums)
        self.x_weighted = None if on_device else self.x / self.norm_factors      # on the device: formed there, fetched by fetch_state()
        self.scale_factors = np.copy(self.norm_factors)
        self.ran_baseline_selection = np.zeros((self.n_local, self.degnorm_iter), dtype=bool)
        self._ran_saved = None
        self._rng = np.random.RandomState(self.random_state)
        self.x_adj = None
        # The O(n p) update between two sweeps runs on the device when it offers it (dn_outer_*): the DI matrix, the
        # weighted and adjusted counts and the flags then stay in HBM and come back once, in fetch_state().
        self._device_outer = self.device_outer and self.n_local > 0 and hasattr(self.dev, 'outer_begin')
        if self._device_outer:
            if on_device:
                self.dev.outer_begin_scaled(self.norm_factors, max(1, self.degnorm_iter))
            else:
                self.dev.outer_begin(self.x_weighted, max(1, self.degnorm_iter))
        self._state_on_device = False
        self.kernel_ms, self.traces, self.class_ms, self.span_ms = [], [], [], []
        self.n_failed, self.n_flagged = [], []
        self.scale_hist, self.rho_raw_hist, self.flags_hist = [], [], []     # per outer iteration: inputs / raw device outputs
        self.offsets_hist = []
        return self.scale_factors

    def _init_failure_text(self, n_bad, n_bad_local, status):
        """What ARPACK would have raised on: count over all ranks, and this rank's first such gene by name."""
        first, noconv = '', False
        if n_bad_local > 0:
            if status is None:                                        # error path only: the statuses were left on the device
                status = self.dev.init_status()
            k = int(np.argmax(status != 0))
            noconv = bool(np.any(status == -4))
            if noconv:
                k = int(np.argmax(status == -4))
            first = ': first {0}'.format(self.gene_names[k] if self.gene_names is not None else 'local gene {0}'.format(k))
        if noconv:
            return ('the rank-1 SVD did not converge within the step cap on {0} gene(s) during initialisation '
                    '(ARPACK would raise ArpackNoConvergence){1}'.format(n_bad, first))
        return 'rank-1 SVD failed on {0} gene(s) during initialisation (all-zero coverage?){1}'.format(n_bad, first)

    def _warn_unconverged(self, i, trace):
        noconv = np.flatnonzero(trace[:, 6] == -4) if trace is not None and len(trace) else np.zeros(0, dtype=int)
        if noconv.size:
            names = [self.gene_names[k] if self.gene_names is not None else 'local gene {0}'.format(k) for k in noconv[:10]]
            logging.warning('DegNorm iteration {0} -- the rank-1 SVD did not converge within the step cap on {1} gene(s) '
                            '(ARPACK would raise ArpackNoConvergence); not used, DI left at 0: {2}{3}'
                            .format(i + 1, noconv.size, ', '.join(names), ' ...' if noconv.size > 10 else ''))

    def _offsets(self, i):
        """
        Systematic-sample starts of this rank's genes for outer iteration i (nmf.py:422).  One stream for the WHOLE data
        set, indexed by global gene id: every rank draws the same n_total numbers and keeps its own, so a sharded run
        is partition-invariant and equals GeneNMFOA.run with the same random_state.
        """
        if self.downsample_rate <= 1:
            return None
        if self.downsample_offsets is not None:
            return np.asarray(self.downsample_offsets[i], dtype=np.int64)
        return self._rng.randint(0, self.downsample_rate, size=self.n_total).astype(np.int64)[self.global_ids]

    def iterate(self, i, want_estimates=False):
        """One outer DegNorm iteration on this rank's genes + the per-sample all-reduce."""
        p = self.p
        ds = self._offsets(i)
        self.offsets_hist.append(ds)
        if self._device_outer:
            return self._iterate_on_device(i, want_estimates, ds)
        if self.n_local > 0:
            rho, flags, trace = self.dev.baseline_iteration(
                self.scale_factors, nmf_iter=self.nmf_iter, bins=self.bins, min_high_coverage=self.min_high_coverage,
                downsample_rate=self.downsample_rate, skip_baseline_selection=self.skip_baseline_selection,
                want_estimates=want_estimates, ds_start=ds)
            self._record_kernel_times()
        else:
            rho, flags, trace = np.zeros((0, p)), np.zeros(0, dtype=bool), np.zeros((0, _lib.TRACE_LEN), dtype=np.int32)
            self.kernel_ms.append(0.0)
            self.class_ms.append((0.0, 0.0, 0.0))
        self.traces.append(trace)
        self._warn_unconverged(i, trace)
        self.scale_hist.append(np.copy(self.scale_factors))
        if self.history_rows is not None:                             # raw device outputs of a few genes (bench.py's parity check)
            self.rho_raw_hist.append(np.copy(rho[self.history_rows]))
            self.flags_hist.append(np.copy(flags[self.history_rows]))
        rho[rho > 0.9] = 0.9                                          # nmf.py:398-399
        rho[rho < 0.] = 0.
        self.ran_baseline_selection[:, i] = flags

        xw = self.x_weighted
        untouched = rho.max(axis=1) == 0 if self.n_local > 0 else np.zeros(0, dtype=bool)   # nmf.py:155
        touched = ~untouched
        A = (xw[touched] / (1 - rho[touched])).sum(axis=0)
        B = xw[untouched].sum(axis=0)
        Wl = xw.sum(axis=0)
        n_fail = float(np.sum(trace[:, 6] != 0)) if trace is not None else 0.0
        n_noconv = float(np.sum(trace[:, 6] == -4)) if trace is not None else 0.0
        avg_di, norm = self._reduce_and_update(i, np.concatenate([A, B, Wl, [float(untouched.sum()), n_fail, n_noconv, float(np.sum(flags))]]))
        if avg_di is not None:
            rho[untouched, :] = avg_di
        self.rho = rho
        self.x_adj = xw / (1 - rho)                                   # nmf.py:581
        self.x_weighted = xw / norm                                   # nmf.py:587
        return self.scale_factors

    def _record_kernel_times(self):
        self.kernel_ms.append(self.dev.last_kernel_ms())
        if hasattr(self.dev, 'class_kernel_ms'):
            self.class_ms.append(tuple(self.dev.class_kernel_ms(c) for c in range(3)))
            self.span_ms.append(self.dev.last_span_ms())

    def _reduce_and_update(self, i, partials, tot=None):
        """
        The per-sample all-reduce of an outer iteration and what every rank derives from it identically
        (nmf.py:148-158, :575-590): returns (avg_di or None, norm factors) and advances the scale factors.
        partials = [A (p), B (p), W (p), #untouched, #failed genes, #unconverged genes, #flagged genes] of this rank.
        """
        p = self.p
        if tot is None:
            tot = self._sum(partials)
        A, B, Wt, n_untouched = tot[:p], tot[p:2 * p], tot[2 * p:3 * p], tot[3 * p]
        self.n_failed.append((int(tot[3 * p + 1]), int(tot[3 * p + 2])))
        self.n_flagged.append(int(tot[3 * p + 3]))                    # ran_baseline_selection[:, i].sum() (nmf.py:571)
        if tot[3 * p + 1] > 0:                                        # the same warning on every rank
            logging.warning('DegNorm iteration {0} -- {1} gene(s) hit a degenerate factorization (the reference would '
                            'raise), {2} of them an eigen-solve that did not converge; their DI scores were left at 0.'
                            .format(i + 1, int(tot[3 * p + 1]), int(tot[3 * p + 2])))
        S_pre = A + B                                                 # colsum of the first x_adj  (nmf.py:575)
        avg_di = None
        if n_untouched > 0:
            avg_di = 1 - (Wt / S_pre)                                 # nmf.py:157
            S_post = A + B / (1 - avg_di)                             # colsum of the second x_adj (nmf.py:581)
        else:
            S_post = S_pre
        self.norm_factors = S_post / np.median(S_post)                # nmf.py:584
        self.scale_factors = self.scale_factors * self.norm_factors   # nmf.py:590
        return avg_di, self.norm_factors

    def _iterate_on_device(self, i, want_estimates, ds):
        """The same iteration with the DI matrix, x_weighted, x_adj and the flags resident in HBM."""
        _, _, trace = self.dev.baseline_iteration(
            self.scale_factors, nmf_iter=self.nmf_iter, bins=self.bins, min_high_coverage=self.min_high_coverage,
            downsample_rate=self.downsample_rate, skip_baseline_selection=self.skip_baseline_selection,
            want_estimates=want_estimates, ds_start=ds, fetch=False,
            trace_out=self._trace_bufs.get(i) if self.reuse_buffers else None)
        if self.reuse_buffers:
            self._trace_bufs[i] = trace
        self._record_kernel_times()
        self.traces.append(trace)
        self.scale_hist.append(np.copy(self.scale_factors))
        if self.history_rows is not None:
            rho_rows, flag_rows = self.dev.fetch_rows(self.history_rows)
            self.rho_raw_hist.append(rho_rows)
            self.flags_hist.append(flag_rows)
        tot = None
        if self._lib_comm:                                            # dn_outer_allreduce: partial sums, RCCL all-reduce, totals -- one call
            tot = self.dev.outer_allreduce()
            self.library_reductions += 1
        elif hasattr(self.comm, 'allreduce_device') and hasattr(self.dev, 'outer_partials_device'):
            ptr, cnt = self.dev.outer_partials_device()               # the sums stay in HBM: the collective works on that buffer
            tot = self.comm.allreduce_device(ptr, cnt, self._dev_id)
        avg_di, norm = self._reduce_and_update(i, self.dev.outer_partials() if tot is None else None, tot)
        if self.n_failed[-1][1] > 0:                                  # some rank's eigen-solve hit the step cap (counted in the all-reduce):
            self._warn_unconverged(i, trace)                          # only then are this rank's counters searched for the genes' names
        self.dev.outer_apply(avg_di, norm, i)
        self._state_on_device = True
        self.rho = self.x_adj = None                                  # stale until fetch_state()
        return self.scale_factors

    def fetch_state(self):
        """Bring rho, x_adj, x_weighted and ran_baseline_selection back from the device (after the last iteration)."""
        if self._state_on_device:
            bufs = self.dev.fetch_outer(self._state_bufs if self.reuse_buffers else None)
            if self.reuse_buffers:
                self._state_bufs = bufs
            self.rho, self.x_adj, self.x_weighted, ran = bufs
            self.ran_baseline_selection = np.array(ran[:, :self.degnorm_iter], dtype=bool)
            if self._ran_saved is not None:                           # the iterations before a re-deal: the device restarted its flags
                k = self._ran_saved.shape[1]
                self.ran_baseline_selection[:, :k] = self._ran_saved
            self._state_on_device = False

    # -- re-dealing the genes from measured cost ------------------------------------------------------
    def redeal(self, class_lengths=None, tol=1.002):
        """
        Level the ranks' loads from what the last outer iteration MEASURED (utils.measured_gene_cost: nmf() calls and active columns
        per gene -- the length alone predicts neither; iteration 1's cost predicts every later one) by moving the few genes that
        change owner (utils.rebalance_moves): their packed float32 coverage, read counts, weighted counts and flags travel point to
        point, every rank re-uploads its new shard and the outer state continues from the weighted counts as they stand, so the
        remaining iterations compute exactly what they would have (per-gene results do not depend on the owner; the per-sample sums
        are added in another order: scale factors agree to round-off).  Collective: every rank calls it after the same iteration.
        The reference deals contiguous equal-count chunks once (nmf_mpi.py:605).  Returns a summary dict (also self.redeal_info).
        """
        comm, size, rank = self.comm, self.comm.size, self.comm.rank
        if size == 1:
            return None
        if self._packed is None:
            raise ValueError('redeal() needs the host copy of the shard (keep_packed = True before load_packed)')
        p, n_tot, done = self.p, self.n_total, len(self.traces)
        if class_lengths is None and self.n_local > 0 and hasattr(self.dev, 'split_length'):
            class_lengths = (self.dev.split_length(), self.dev.tiny_length())
        cost_l = measured_gene_cost(self.traces[-1], self._lengths, p, class_lengths, self.downsample_rate) if self.n_local > 0 else np.zeros(0)
        dense = np.zeros(3 * n_tot)
        dense[self.global_ids] = cost_l
        dense[n_tot + self.global_ids] = rank + 1.0
        dense[2 * n_tot + self.global_ids] = self._lengths
        dense = _allreduce(comm, dense)                               # every rank: all costs, owners, lengths (a setup collective, host path)
        cost, owner, lens = dense[:n_tot], np.rint(dense[n_tot:2 * n_tot]).astype(np.int64) - 1, np.rint(dense[2 * n_tot:]).astype(np.int64)
        load0 = np.bincount(owner, weights=cost, minlength=size)
        moves = rebalance_moves(owner, cost, size, tol=tol)
        # the state that travels with a gene
        if self._device_outer and self._state_on_device:
            self.fetch_state()
        ran = np.asarray(self.ran_baseline_selection, dtype=np.uint8).reshape(self.n_local, -1)[:, :done]
        offs = np.zeros(self.n_local + 1, dtype=np.int64)
        np.cumsum(p * self._lengths, out=offs[1:])
        pos = {int(g): k for k, g in enumerate(self.global_ids)}
        pairs = OrderedDict()
        for g, src, dst in moves:
            pairs.setdefault((src, dst), []).append(g)
        incoming = []
        for (src, dst), genes in sorted(pairs.items()):              # one global order of two-rank exchanges: no cycle of waits
            genes = sorted(genes)
            if rank == src:
                ks = [pos[g] for g in genes]
                cov = np.concatenate([self._packed[offs[k]:offs[k + 1]] for k in ks])
                _send_arrays(comm, dst, 1200 + src, (cov, self.x[ks], np.asarray(self.x_weighted)[ks], ran[ks]))
            elif rank == dst:
                n_val = int(p * lens[genes].sum())
                cov, xr, xw, rn = _recv_arrays(comm, src, 1200 + src, [((n_val,), np.float32), ((len(genes), p), np.float64),
                                                                      ((len(genes), p), np.float64), ((len(genes), done), np.uint8)])
                incoming.append((genes, cov, xr, xw, rn))
        gone = set(g for g, src, dst in moves if src == rank)
        keep = [k for k, g in enumerate(self.global_ids) if int(g) not in gone]
        rows = [(int(self.global_ids[k]), self._packed[offs[k]:offs[k + 1]], self.x[k], np.asarray(self.x_weighted)[k], ran[k]) for k in keep]
        for genes, cov, xr, xw, rn in incoming:
            o = 0
            for j, g in enumerate(genes):
                n_val = int(p * lens[g])
                rows.append((int(g), cov[o:o + n_val], xr[j], xw[j], rn[j]))
                o += n_val
        rows.sort(key=lambda r: r[0])
        ids = np.array([r[0] for r in rows], dtype=np.int64)
        new_packed = np.concatenate([r[1] for r in rows]) if rows else np.zeros(0, dtype=np.float32)
        new_x = np.array([r[2] for r in rows], dtype=np.float64).reshape(len(rows), p)
        new_xw = np.array([r[3] for r in rows], dtype=np.float64).reshape(len(rows), p)
        new_ran = np.array([r[4] for r in rows], dtype=np.uint8).reshape(len(rows), done)
        names = None
        if self.gene_names is not None:                               # names travel as a small object per pair
            names = dict(zip((int(g) for g in self.global_ids), self.gene_names))
            for (src, dst), genes in sorted(pairs.items()):
                if rank == src:
                    comm.send([names[g] for g in sorted(genes)], dest=dst, tag=1300 + src)
                elif rank == dst:
                    names.update(zip(sorted(genes), comm.recv(source=src, tag=1300 + src)))
        # the new shard: upload, read counts, weighted counts; the iterations continue
        scale, norm = np.copy(self.scale_factors), np.copy(self.norm_factors)
        keep_traces, keep_hist = self.traces, (self.scale_hist, self.n_failed, self.n_flagged, self.kernel_ms, self.class_ms, self.span_ms, self.offsets_hist)
        self._trace_bufs, self._state_bufs = {}, None
        self.load_packed(new_packed, lens[ids], p, new_x, global_ids=ids, n_total=n_tot)
        self.x_weighted = new_xw
        self.ran_baseline_selection = np.zeros((self.n_local, self.degnorm_iter), dtype=bool)
        self.ran_baseline_selection[:, :done] = new_ran.astype(bool)
        self._ran_saved = new_ran.astype(bool)
        self.scale_factors, self.norm_factors = scale, norm
        self.gene_names = [names[int(g)] for g in ids] if names is not None else None
        self._device_outer = self.device_outer and self.n_local > 0 and hasattr(self.dev, 'outer_begin')
        if self._device_outer:
            self.dev.outer_begin(new_xw, max(1, self.degnorm_iter))
        self._state_on_device = False
        self.traces = []                                              # per-gene rows of earlier iterations belong to another gene set
        (self.scale_hist, self.n_failed, self.n_flagged, self.kernel_ms, self.class_ms, self.span_ms, self.offsets_hist) = keep_hist
        self._traces_before_redeal = keep_traces
        owner2 = owner.copy()
        for g, src, dst in moves:
            owner2[g] = dst
        load1 = np.bincount(owner2, weights=cost, minlength=size)
        self.owner = owner2
        self.redeal_info = {'moves': len(moves), 'after_iteration': done, 'max_over_mean_before': float(load0.max() / load0.mean()),
                            'max_over_mean_after': float(load1.max() / load1.mean()), 'genes_per_rank': np.bincount(owner2, minlength=size).tolist(),
                            'bytes_moved': int(4 * p * sum(int(lens[g]) for g, _, _ in moves))}
        return self.redeal_info


    def run(self, want_estimates=True, flat=False):
        """Returns the estimates of the last iteration: a list of (p x L) arrays, or (flat buffer, lengths) if `flat`."""
        self.initialize()
        est = None
        for i in range(self.degnorm_iter):
            last = i == self.degnorm_iter - 1
            self.iterate(i, want_estimates=want_estimates and last)
            if i == 0 and self.redeal_after_first and not last and self.comm.size > 1:
                self.redeal()
            if want_estimates and last:
                if self.n_local == 0:
                    est = (np.zeros(0), np.zeros(0, dtype=np.int64)) if flat else []
                elif flat:
                    buf = (self.dev.fetch_estimates_flat() if hasattr(self.dev, 'fetch_estimates_flat')
                           else np.concatenate([m.ravel() for m in self.dev.fetch_estimates()]))
                    est = (buf, np.asarray(self.dev.lengths, dtype=np.int64))
                else:
                    est = self.dev.fetch_estimates()
        self.fetch_state()
        return est


# ----------------------------------------------------------------------------------------------- #
def _pack_f32(mats, n_threads=0):
    """(p x L_g) matrices -> (packed float32, lengths int64, number of values float32 cannot hold exactly).  The conversion runs on
    host threads (numpy releases the GIL in the copy): rank 0 packs every share of the data set before it can ship it."""
    lengths = np.array([m.shape[1] for m in mats], dtype=np.int64)
    if len(mats) == 0:
        return np.zeros(0, dtype=np.float32), lengths, 0
    p = int(mats[0].shape[0])
    offs = np.zeros(len(mats) + 1, dtype=np.int64)
    np.cumsum(p * lengths, out=offs[1:])
    packed = np.empty(int(offs[-1]), dtype=np.float32)
    n_threads = int(n_threads) if n_threads else max(1, min(16, os.cpu_count() or 1))

    def work(lo_hi):
        bad = 0
        for k in range(*lo_hi):
            m = np.asarray(mats[k])
            dst = packed[offs[k]:offs[k + 1]].reshape(m.shape)
            np.copyto(dst, m, casting='unsafe')
            if m.dtype != np.float32:
                bad += int(np.count_nonzero(dst != m))
        return bad
    step = max(1, (len(mats) + 4 * n_threads - 1) // (4 * n_threads))
    chunks = [(lo, min(len(mats), lo + step)) for lo in range(0, len(mats), step)]
    if n_threads > 1 and len(mats) > 64:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=n_threads) as ex:
            inexact = sum(ex.map(work, chunks))
    else:
        inexact = sum(work(c) for c in chunks)
    return packed, lengths, int(inexact)


def _class_lengths(device, p, downsample_rate):
    """(split_len, tiny_len) of a p-sample cohort on this rank's GPU, or None when the device cannot say (test stand-ins)."""
    try:
        dev = _lib.Device(int(os.environ.get('LOCAL_RANK', 0)) if device is None else int(device))
    except Exception:
        return None
    try:
        return dev.class_lengths(p, downsample_rate) if hasattr(dev, 'class_lengths') else None
    finally:
        dev.close()


def _partition(li_vec, size, p, downsample_rate, partition, device):
    if partition == 'contiguous':
        parts = split_into_chunks(list(range(len(li_vec))), size)      # nmf_mpi.py:605
    elif partition == 'balanced':
        parts = partition_by_cost(li_vec, size, p=p, downsample_rate=abs(int(downsample_rate)),
                                  class_lengths=_class_lengths(device, p, abs(int(downsample_rate))))
    else:
        raise ValueError("partition must be 'balanced' or 'contiguous'")
    parts = [list(q) for q in parts]
    while len(parts) < size:                                            # fewer chunks than ranks: idle ranks get nothing
        parts.append([])
    return parts


def _run_shard_and_gather(comm, eng_kw, my_names, packed, lengths, my_x, my_ids, n_genes, p, degnorm_iter,
                          all_genes, li_vec, parts, want_estimates=True, timings=None, redeal=False):
    """
    Every rank: its resident shard through the engine; then rank 0 collects -- and ONLY rank 0: raw buffers point to point, in
    rank order (nmf_mpi.py:796-815 gathers the same things as pickled tuples), rows scattered back into the original gene order.
    `all_genes`, `li_vec`, `parts` are needed on rank 0 only (the receiver knows every shape from them).
    """
    import time
    size, rank = comm.size, comm.rank
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    eng = ShardedNMFOA(comm=comm, **eng_kw)
    eng.gene_names = my_names
    eng.keep_packed = eng.redeal_after_first = bool(redeal) and size > 1 and abs(int(degnorm_iter)) > 1
    eng.load_packed(packed, lengths, p, my_x, global_ids=my_ids, n_total=n_genes)
    del packed
    tm['upload_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    flat, lens = eng.run(want_estimates=want_estimates, flat=True) if want_estimates else (eng.run(want_estimates=False), None)
    if not want_estimates:
        flat = np.zeros(0)
    tm['run_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    logging.info('({0}/{1}) -- finished {2} genes'.format(rank + 1, size, eng.n_local))
    if eng.redeal_info is not None:                                     # the genes changed hands after the first iteration: every rank knows the owners
        parts = [np.flatnonzero(eng.owner == r).tolist() for r in range(size)]
        tm['redeal'] = eng.redeal_info
    n_it = abs(int(degnorm_iter))
    mine = (flat, np.ascontiguousarray(eng.rho, dtype=np.float64).reshape(eng.n_local, p),
            np.ascontiguousarray(eng.x_adj, dtype=np.float64).reshape(eng.n_local, p),
            np.ascontiguousarray(eng.ran_baseline_selection, dtype=np.uint8).reshape(eng.n_local, n_it))
    if rank != 0:
        _send_arrays(comm, 0, 666 + rank, mine)
        comm.Barrier()
        tm['gather_s'] = time.perf_counter() - t0
        return None
    li_vec = np.asarray(li_vec, dtype=np.int64)
    rho, x_adj = np.empty((n_genes, p)), np.empty((n_genes, p))
    ran = np.empty((n_genes, n_it), dtype=bool)
    flats = [None] * size
    for r in range(size):
        idx = np.asarray(parts[r], dtype=np.int64)
        if r == 0:
            got = mine
        else:
            n_r = len(idx)
            n_est = int(p * li_vec[idx].sum()) if want_estimates else 0
            got = _recv_arrays(comm, r, 666 + r, [((n_est,), np.float64), ((n_r, p), np.float64), ((n_r, p), np.float64), ((n_r, n_it), np.uint8)])
        flats[r] = got[0]
        if len(idx):
            rho[idx], x_adj[idx], ran[idx] = got[1], got[2], got[3].astype(bool)
    comm.Barrier()
    tm['gather_s'] = time.perf_counter() - t0
    estimates = None
    if want_estimates:
        by_pos = [None] * n_genes
        for r in range(size):
            o = 0
            for k in parts[r]:
                L = int(li_vec[k])
                by_pos[k] = flats[r][o:o + p * L].reshape(p, L)         # views of the rank's buffer: no further copy
                o += p * L
        estimates = OrderedDict((g, by_pos[k]) for k, g in enumerate(all_genes))      # original gene order (nmf_mpi.py:852-860)
    return {'estimates': estimates, 'rho': rho, 'x_adj': x_adj, 'ran_baseline_selection': ran}


def run_gene_nmfoa_mpi(comm, cov_dat, reads_dat, degnorm_iter=5, downsample_rate=1, min_high_coverage=50,
                       nmf_iter=100, bins=20, n_jobs=1, skip_baseline_selection=False, random_state=123,
                       device=None, partition='balanced', want_estimates=True, timings=None, redeal=False):
    """
    Gene-sharded DegNorm run with the reference's signature (nmf_mpi.py:555-580).  Rank 0 holds
    ``cov_dat`` (OrderedDict gene -> p x L) and ``reads_dat`` (n x p) and ships each worker its share once
    (nmf_mpi.py:603-629) -- as ONE packed float32 buffer per worker plus three small arrays, sent as raw bytes (the layout the
    device consumes: half the bytes of the reference's float64 pickles, no per-gene objects, no pickling of the data); every rank
    then keeps its genes on its own GPU.  Results come back to rank 0 only, as raw buffers (round 3 all-gathered pickles: every
    rank received every rank's estimates).
    ``redeal`` (extra): after the first outer iteration the ranks level their MEASURED per-gene cost by handing over the few genes
    that change owner (ShardedNMFOA.redeal: the length predicts a gene's cost poorly, its first iteration predicts the others exactly).
    ``partition`` (extra): 'balanced' deals the genes by predicted cost so that every GPU gets the same share of every gene
    class (utils.partition_by_cost with the class boundaries of THIS sample count on the device, dn_class_lengths),
    'contiguous' is the reference's equal-count chunking (nmf_mpi.py:605); per-gene results do not depend on it (the
    down-sampling offsets are drawn per global gene id), rows come back in the original order.  ``want_estimates`` (extra):
    False leaves the p x L estimates on the GPUs ('estimates': None) for callers that only need the DI scores.
    Input errors found on rank 0 are raised on EVERY rank (nobody is left waiting in a receive), and a rank without genes
    (fewer chunks than ranks, nmf_mpi.py:613) takes part in the collectives with zeros.
    Returns, on rank 0, {'estimates': {gene: p x L}, 'rho', 'x_adj', 'ran_baseline_selection'} in the
    original gene order (nmf_mpi.py:852-860), None elsewhere.
    """
    import time
    size, rank = comm.size, comm.rank
    t_start = time.perf_counter()
    err, parts, n_genes, p, all_genes, li_vec = None, None, 0, 0, None, None
    if rank == 0:
        try:
            all_genes = list(cov_dat.keys())
            n_genes = len(all_genes)
            x = np.array(reads_dat, dtype=np.float64)
            if n_genes == 0:
                raise ValueError('no coverage matrices')
            if x.shape[0] != n_genes:
                raise ValueError('Number of genes in read count matrix not equal to number of coverage matrices!')
            if not all(getattr(z, 'ndim', 0) == 2 for z in cov_dat.values()):
                raise ValueError('Not all coverage matrices are 2-d arrays!')
            li_vec = np.array([z.shape[1] for z in cov_dat.values()])
            p = int(next(iter(cov_dat.values())).shape[0])
            if not all(z.shape[0] == p for z in cov_dat.values()):
                raise ValueError('coverage matrices disagree on the number of samples')
            if x.ndim != 2 or x.shape[1] != p:
                raise ValueError('read count matrix must be (number of genes) x (number of samples)')
            if abs(int(downsample_rate)) > 1 and not np.min(li_vec) >= abs(int(downsample_rate)):
                raise ValueError('downsample_rate is too large; take-every size > at least one gene.')
            parts = _partition(li_vec, size, p, downsample_rate, partition, device)
        except Exception as e:                                            # ANY failure of the checks reaches every rank: nobody waits in a receive
            err = str(e) if isinstance(e, ValueError) else '{0}: {1}'.format(type(e).__name__, e)
    err, n_genes, p = _bcast(comm, (err, n_genes, p))
    if err is not None:
        raise ValueError(err)

    if rank == 0:
        n_inexact = 0
        pending = []
        for r in range(size - 1, -1, -1):                                # own share last; a share travels while the next one is packed
            idx = parts[r]
            packed, lengths, bad = _pack_f32([cov_dat[all_genes[k]] for k in idx])
            n_inexact += bad
            if r > 0:
                comm.send(([all_genes[k] for k in idx], int(packed.size)), dest=r, tag=333 + r)      # names + sizes: the only pickle
                _send_arrays(comm, r, 444 + r, (packed, lengths, x[idx], np.asarray(idx, dtype=np.int64)), pending)
        if pending:
            comm.wait_sends(pending)
        if n_inexact:
            logging.warning('{0} coverage values are not exactly representable in float32; they were rounded on upload.'.format(n_inexact))
        my_names, my_x, my_ids = [all_genes[k] for k in parts[0]], x[parts[0]], np.asarray(parts[0], dtype=np.int64)
    else:
        my_names, n_val = comm.recv(source=0, tag=333 + rank)
        n_r = len(my_names)
        packed, lengths, my_x, my_ids = _recv_arrays(comm, 0, 444 + rank, [((n_val,), np.float32), ((n_r,), np.int64),
                                                                           ((n_r, p), np.float64), ((n_r,), np.int64)])
    if timings is not None:
        timings['scatter_s'] = time.perf_counter() - t_start              # checks, partition, packing, shipping the shares
    eng_kw = dict(device=device, degnorm_iter=degnorm_iter, downsample_rate=downsample_rate, min_high_coverage=min_high_coverage,
                  nmf_iter=nmf_iter, bins=bins, skip_baseline_selection=skip_baseline_selection, random_state=random_state)
    return _run_shard_and_gather(comm, eng_kw, my_names, packed, lengths, my_x, my_ids, n_genes, p, degnorm_iter,
                                 all_genes, li_vec, parts, want_estimates=want_estimates, timings=timings, redeal=redeal)


def save_results(genes_df, estimates, rho, x_adj, ran_baseline_selection, sample_ids=None, output_dir='.'):
    """Module-level writer with the reference's signature (nmf_mpi.py:448-552; called at __main_mpi__.py:450-456)."""
    genes = list(estimates.keys())
    write_results(genes=genes, estimates=estimates, rho=rho, x_adj=x_adj,
                  ran_baseline_selection=ran_baseline_selection, gene_manifest_df=genes_df,
                  output_dir=output_dir, sample_ids=sample_ids)
