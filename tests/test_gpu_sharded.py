"""
The gene-sharded driver end to end on real HIP devices: two processes (one torch.distributed rank each, gloo for the
collective because the test box has ONE GPU -- with one GPU per rank the same code runs over nccl = RCCL), both on
device 0, against the reference's own MPI outputs (tests/golden/mpi.npz).  What test_sharded_cpu.py checks with the
oracle standing in for the device, this checks with the kernels.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import golden                       # noqa: E402
from degnorm_amd import synth                     # noqa: E402

pytestmark = pytest.mark.gpu


def _inputs():
    G = golden('mpi')
    covs = [synth.synth_gene(int(G['seed']), int(g), int(G['p']), int(G['l_min']), int(G['l_max']))[0]
            for g in G['gene_ids']]
    cov_dat = OrderedDict(('gene_%06d' % g, c) for g, c in zip(G['gene_ids'], covs))
    return G, cov_dat


def _worker(rank, world, port, partition, out_q, redeal=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    from degnorm_amd.nmf_mpi import run_gene_nmfoa_mpi, TorchComm
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        G, cov_dat = _inputs()
        comm = TorchComm()
        tm = {}
        res = run_gene_nmfoa_mpi(comm, cov_dat if rank == 0 else None, G['reads'] if rank == 0 else None,
                                 degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']), device=0,
                                 partition=partition, redeal=redeal, timings=tm)
        if rank == 0:
            out = {k: (v if k != 'estimates' else {g: e.sum(axis=1) for g, e in v.items()}) for k, v in res.items()}
            out['redeal'] = tm.get('redeal')
            out_q.put(out)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('partition', ['balanced', 'contiguous'])
def test_two_ranks_on_the_gpu_match_reference_mpi_golden(partition):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, partition, out_q)) for r in range(2)]
    [p.start() for p in procs]
    res = out_q.get(timeout=300)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    G, cov_dat = _inputs()
    np.testing.assert_allclose(res['rho'], G['mpi2_rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res['x_adj'], G['mpi2_x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(res['ran_baseline_selection'], G['mpi2_flags'])
    assert list(res['estimates'].keys()) == list(cov_dat.keys())
    np.testing.assert_allclose(np.vstack(list(res['estimates'].values())), G['mpi2_est_rowsum'], rtol=1e-9)


def test_two_ranks_on_the_gpu_redeal_after_first_iteration():
    """
    ShardedNMFOA.redeal on real devices (both ranks on GPU 0 over gloo): the contiguous deal of the reference (nmf_mpi.py:605) is
    levelled after the first outer iteration from the kernels' own counters -- coverage re-uploaded, the device-side outer state
    restarted from the weighted counts as they stand -- and the run still gives the reference's MPI result.
    """
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    port = 29500 + ((os.getpid() + 13) % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 'contiguous', out_q, True)) for r in range(2)]
    [p.start() for p in procs]
    res = out_q.get(timeout=300)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    G, cov_dat = _inputs()
    assert res['redeal'] is not None and res['redeal']['after_iteration'] == 1
    assert res['redeal']['max_over_mean_after'] <= res['redeal']['max_over_mean_before'] + 1e-12
    np.testing.assert_allclose(res['rho'], G['mpi2_rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res['x_adj'], G['mpi2_x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(res['ran_baseline_selection'], G['mpi2_flags'])
    assert list(res['estimates'].keys()) == list(cov_dat.keys())
    np.testing.assert_allclose(np.vstack(list(res['estimates'].values())), G['mpi2_est_rowsum'], rtol=1e-9)


def _nccl_worker(port, out_q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import torch
    import torch.distributed as dist
    from degnorm_amd.nmf_mpi import ShardedNMFOA, TorchComm
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        G, cov_dat = _inputs()
        comm = TorchComm(device='cuda:0')
        assert comm.backend == 'nccl' and comm.size == 1
        probe = comm.allreduce_sum(np.arange(31, dtype=np.float64))            # the 3p+1-sized collective itself, on the device
        eng = ShardedNMFOA(comm=comm, device=0, degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']))
        eng.load(list(cov_dat.values()), G['reads'])
        eng.run(want_estimates=False)
        comm.Barrier()
        out_q.put(dict(rho=eng.rho, x_adj=eng.x_adj, flags=eng.ran_baseline_selection, probe=probe,
                       rccl='.'.join(str(v) for v in torch.cuda.nccl.version()),
                       device_reductions=getattr(comm, 'device_reductions', 0), n_flagged=list(eng.n_flagged)))
    finally:
        dist.destroy_process_group()


def test_sharded_driver_over_rccl_world_size_1():
    """
    The collective path as the multi-GPU bench runs it -- torch.distributed backend "nccl" (= RCCL on ROCm), float64
    tensors on the device, TorchComm.allreduce_sum -- at the one world size this box offers; results == the reference's
    single-node rows of tests/golden/mpi.npz (run_gene_nmfoa_mpi == GeneNMFOA.run there).  Own process: the process group
    must not outlive the test.
    """
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    proc = ctx.Process(target=_nccl_worker, args=(29500 + ((os.getpid() + 7) % 2000), out_q))
    proc.start()
    res = out_q.get(timeout=300)
    proc.join(timeout=60)
    assert proc.exitcode == 0
    G, _ = _inputs()
    np.testing.assert_array_equal(res['probe'], np.arange(31, dtype=np.float64))
    np.testing.assert_allclose(res['rho'], G['single_rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res['x_adj'], G['single_x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(res['flags'], G['single_flags'])
    assert res['rccl']
    # the per-iteration collective ran IN PLACE on the library's device buffer (dn_outer_partials_device), once per iteration,
    # and carried the count of genes sent through baseline selection
    assert res['device_reductions'] == int(G['degnorm_iter'])
    np.testing.assert_array_equal(res['n_flagged'], G['single_flags'].sum(axis=0))


def test_outer_update_on_the_device_equals_the_host_update():
    """
    The O(n p) arithmetic between two sweeps (DI clip, correct_di_scores, x_adj, normalisation: nmf.py:398-399, :148-158,
    :575-590) on the device (dn_outer_*: the DI matrix never leaves HBM) against the same in numpy on the host, and both
    against the reference's golden run on config-2 genes (3 outer iterations; untouched genes occur).
    """
    from degnorm_amd.nmf_mpi import ShardedNMFOA
    G = golden('run_c2')
    covs = [synth.synth_gene(int(G['seed']), int(g), int(G['p']), int(G['l_min']), int(G['l_max']))[0] for g in G['gene_ids']]
    out = []
    for on_device in (True, False):
        eng = ShardedNMFOA(device=0, degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']))
        eng.device_outer = on_device
        eng.load(covs, G['reads'])
        eng.run(want_estimates=False)
        assert eng._device_outer == on_device
        out.append(eng)
        np.testing.assert_allclose(eng.rho, G['rho'], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(eng.x_adj, G['x_adj'], rtol=1e-9)
        np.testing.assert_allclose(eng.x_weighted, G['x_weighted'], rtol=1e-9)
        np.testing.assert_allclose(eng.scale_factors, G['scale_factors'], rtol=1e-9)
        np.testing.assert_array_equal(eng.ran_baseline_selection, G['ran_baseline_selection'])
    np.testing.assert_allclose(out[0].rho, out[1].rho, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out[0].x_adj, out[1].x_adj, rtol=1e-12)
    assert (out[0].rho.max(axis=1) == 0).sum() == 0 and (G['rho_hist'][0].max(axis=1) == 0).any()     # untouched genes were corrected


def _libcomm_worker(out_q):
    # no torch in this process: the library binds librccl by itself (dlopen), the communicator lives in the handle
    from degnorm_amd.nmf_mpi import ShardedNMFOA, LocalComm
    from degnorm_amd import _lib
    G, cov_dat = _inputs()
    eng = ShardedNMFOA(comm=LocalComm(), device=0, degnorm_iter=int(G['degnorm_iter']), nmf_iter=int(G['nmf_iter']))
    eng.load(list(cov_dat.values()), G['reads'])
    eng.attach_library_comm()
    probe = eng.dev.comm_allreduce(np.arange(34, dtype=np.float64))
    eng.run(want_estimates=False)
    out_q.put(dict(rho=eng.rho, x_adj=eng.x_adj, flags=eng.ran_baseline_selection, probe=probe, lib=_lib.Device.comm_library(),
                   size=eng.dev.comm_size(), reductions=eng.library_reductions, n_flagged=list(eng.n_flagged),
                   torch_loaded='torch' in sys.modules))
    eng.dev.comm_destroy()


def test_collective_inside_the_library_world_size_1():
    """
    include/degnorm_amd.h dn_comm_*: the sharded run's collective behind the C ABI -- dn_comm_unique_id / dn_comm_create
    (ncclCommInitRank on the handle's GPU), dn_init_allreduce and dn_outer_allreduce (partial sums -> ncclAllReduce in place on the
    library's stream -> totals), with NO torch in the process -- at the one world size this box offers, against the reference's rows
    of tests/golden/mpi.npz.  One reduction for the initial normalisation and one per outer iteration.
    """
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    proc = ctx.Process(target=_libcomm_worker, args=(out_q,))
    proc.start()
    res = out_q.get(timeout=300)
    proc.join(timeout=60)
    assert proc.exitcode == 0
    G, _ = _inputs()
    np.testing.assert_array_equal(res['probe'], np.arange(34, dtype=np.float64))
    np.testing.assert_allclose(res['rho'], G['single_rho'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res['x_adj'], G['single_x_adj'], rtol=1e-9)
    np.testing.assert_array_equal(res['flags'], G['single_flags'])
    assert 'rccl' in res['lib'] and res['size'] == 1
    assert res['reductions'] == 1 + int(G['degnorm_iter'])
    np.testing.assert_array_equal(res['n_flagged'], G['single_flags'].sum(axis=0))
