"""
bench.py -- DegNorm NMF-OA hot path on MI355X: genes/sec on BASELINE.json's config 2
(20 000 synthetic genes x 10 samples, L ~ U[200, 5000], 5 DegNorm iterations, nmf_iter = 100).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genes G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete DegNorm run over the resident genes: the ratio-SVD initialisation pass plus
`--iters` (5) outer iterations, each = one launch of the baseline-selection kernel over every gene of the
rank's shard, the D2H of the DI rows / flags / traces and the per-sample all-reduce (RCCL over xGMI for
N > 1).  Coverage is generated and uploaded to HBM before the timed region (estimates are not fetched).
The 20 000 genes are sharded across the N ranks (contiguous chunks like nmf_mpi.py:605), so scaling is
strong; `value` = total genes / max-over-ranks wall time.

Rank 0 prints ONE JSON line with the driver contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # AMD MI355X spec, fp64 vector (= 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz); not in the guide


def algorithmic_bytes(trace, lengths, p, nmf_iter, mask=None):
    """
    SURVEY.md 8(d): per gene and outer iteration, fp32 storage, one fused pass per inner NMF-OA iteration that
    reads x and lambda and writes lambda:  bytes_g = 4 p [ L_g + sum_k n_{g,k} (3T + 3) + L_g ],
    with sum_k n_{g,k} (active columns summed over the gene's nmf() calls) taken from the device counters.
    `mask` selects the genes one kernel processes (the two gene classes run in separate launches).
    """
    sum_cols = trace[:, 2].astype(np.float64)
    per_gene = 4.0 * p * (2.0 * lengths + sum_cols * (3.0 * nmf_iter + 3.0))
    return float(per_gene[mask].sum() if mask is not None else per_gene.sum())


def pmc_traffic(kernel_name, genes_in_kernel):
    """
    HBM bytes per k_baseline launch from the PMC counters.  bench.py cannot collect PMCs itself: the figure comes
    from the committed rocprofv3 passes of THIS command (profiles/round1/pmc_traffic.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 calibration) and is only reported when the
    profiled kernel (name and gene class) is the one that just ran.
    """
    path = os.path.join(ROOT, 'profiles', 'round1', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get('kernel') == kernel_name and int(d.get('genes_in_kernel', -1)) == int(genes_in_kernel):
            return float(d['hbm_bytes_per_launch'])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(cfg, p, nmf_iter, iters, n_sample):
    """The CPU oracle (oracle/, parity-pinned port of the reference) timed on this box's host cores."""
    from oracle import oracle as orc
    from degnorm_amd import synth
    orc.build()
    cores = int(orc.lib().dno_max_threads())
    try:
        cores = max(1, min(cores, len(os.sched_getaffinity(0))))      # the CPUs this process may run on ...
    except AttributeError:
        pass
    try:                                                                # ... capped by the cgroup CPU quota (GPU box: 16)
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    covs = [synth.synth_gene(cfg['seed'], g, p, cfg['l_min'], cfg['l_max'])[0] for g in range(n_sample)]
    reads = np.vstack([synth.read_counts_from_coverage(c) for c in covs])
    t0 = time.time()
    orc.run(covs, reads, degnorm_iter=iters, nmf_iter=nmf_iter, n_threads=cores)
    dt = time.time() - t0
    return {'value': n_sample / dt, 'unit': 'genes/sec', 'cores': cores, 'kind': 'port',
            'sample': 'first {0} genes of the config-2 generator, full run ({1} outer iterations, nmf_iter {2}), '
                      'oracle/nmfoa_oracle.c with OpenMP over genes, {3:.1f} s wall'.format(n_sample, iters, nmf_iter, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--genes', type=int, default=20000, help='total genes (config 2: 20000)')
    ap.add_argument('--iters', type=int, default=5, help='outer DegNorm iterations per step')
    ap.add_argument('--nmf-iter', type=int, default=100)
    ap.add_argument('--cpu-sample', type=int, default=768, help='genes in the CPU-baseline sample (0 = skip)')
    ap.add_argument('--warmup-genes', type=int, default=0, help='0: warm up on the full shard')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
        args.gpus = world

    import torch
    from degnorm_amd import synth
    from degnorm_amd.nmf_mpi import ShardedNMFOA, TorchComm, LocalComm
    from degnorm_amd.utils import partition_by_length

    torch.cuda.set_device(local_rank)
    comm = LocalComm()
    distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ     # under torchrun use RCCL even at N = 1
    if distributed:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        comm = TorchComm(device='cuda:{0}'.format(local_rank))

    cfg = dict(synth.CONFIGS['c2'])
    p = cfg['p']
    # every rank gets the same total gene length (the generator's lengths are a cheap pure function of the gene id)
    all_len = [synth.gene_length(cfg['seed'], g, cfg['l_min'], cfg['l_max']) for g in range(args.genes)] if world > 1 else None
    parts = partition_by_length(all_len, world) if world > 1 else [list(range(args.genes))]
    my_genes = parts[rank] if rank < len(parts) else []
    t_gen = time.time()
    packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], my_genes, p, cfg['l_min'], cfg['l_max'],
                                                   n_threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    t_gen = time.time() - t_gen

    eng = ShardedNMFOA(comm=comm, device=local_rank, degnorm_iter=args.iters, nmf_iter=args.nmf_iter)
    t_up = time.time()
    eng.load_packed(packed, lengths, p, reads)
    t_up = time.time() - t_up

    def sync():
        torch.cuda.synchronize()
        comm.Barrier()
        torch.cuda.synchronize()

    def step():
        eng.initialize()
        for i in range(args.iters):
            eng.iterate(i, want_estimates=False)

    for _ in range(args.warmup):
        step()

    # dominant kernel = the wide-gene class (genes longer than the split length, one 256-thread workgroup per CU)
    split = eng.dev.split_length()
    wide = lengths > split if split > 0 else np.ones(len(lengths), dtype=bool)
    kernel_ms, alg_bytes, alg_all, narrow_ms = [], [], [], []
    sync()
    t0 = time.time()
    all_traces = []
    for _ in range(args.steps):
        step()
        kernel_ms += [c[0] for c in eng.class_ms]
        narrow_ms += [c[1] for c in eng.class_ms]
        all_traces += eng.traces                                       # accounting happens after the clock stops
    sync()
    dt = time.time() - t0
    alg_bytes = [algorithmic_bytes(tr, lengths, p, args.nmf_iter, wide) for tr in all_traces]
    alg_all = [algorithmic_bytes(tr, lengths, p, args.nmf_iter) for tr in all_traces]

    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = args.genes * args.steps / dt
        avg_ms = float(np.mean(kernel_ms))
        avg_bytes = float(np.mean(alg_bytes))
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = pmc_traffic(eng.dev.class_kernel_name(0), int(wide.sum())) if (world == 1 and args.genes == 20000) else None
        # fp64 vector work of the inner passes: per column and inner iteration u.a (2p), the update (5p), the Gram
        # update (p(p+1)) and the 1/s scaling (p) -- the unit that actually bounds the kernel ("What bounds it")
        col_iters = float(np.mean([float(tr[wide, 2].astype(np.float64).sum()) * args.nmf_iter for tr in all_traces]))
        flop = col_iters * (p * p + 9.0 * p)
        out = {
            'metric': 'genes/sec (20k genes x 10 samples, 5 iters)',
            'value': value, 'unit': 'genes/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config 2: {0} synthetic genes x {1} samples, L~U[{2},{3}], {4} DegNorm iters, '
                                   'nmf_iter {5}, fp32 coverage in HBM, fp64 arithmetic'
                                   .format(args.genes, p, cfg['l_min'], cfg['l_max'], args.iters, args.nmf_iter),
                       'genes_per_gpu': len(my_genes), 'sharding': 'length-balanced gene partition, 1 all-reduce of 3p+1 f64 per outer iter'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': traffic,
                         'traffic_rate_gbps': (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                         'traffic_frac': (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         'fp64_valu': {'achieved': flop / (avg_ms * 1e-3) / 1e12, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                       'frac': flop / (avg_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                       'flop_per_column_iteration': p * p + 9.0 * p},
                         'kernel': eng.dev.class_kernel_name(0), 'avg_launch_ms': avg_ms,
                         'algorithmic_bytes_per_launch': avg_bytes, 'launches_timed': len(kernel_ms),
                         'genes_in_kernel': int(wide.sum()), 'split_length': split,
                         'second_kernel': {'kernel': eng.dev.class_kernel_name(1), 'genes': int((~wide).sum()),
                                           'algorithmic_bytes_per_launch': float(np.mean(alg_all)) - avg_bytes,
                                           'launch_to_end_ms': float(np.mean(narrow_ms)),
                                           'note': 'narrow genes, 128-thread workgroups, fills CUs as the wide class drains'},
                         'note': 'rank-0 shard; HIP events on the library stream around each launch.  achieved = algorithmic '
                                 'bytes of SURVEY 8(d) (fp32 x and lambda re-streamed every inner iteration) / time: the kernel keeps '
                                 'x + lambda of the first ~2000 columns of a gene in LDS, so this figure can exceed the HBM peak; '
                                 'traffic* = what the fabric-side counters saw'},
            'setup': {'synth_s': t_gen, 'upload_s': t_up},
        }
        if world == 1 and args.cpu_sample > 0:
            out['cpu_baseline'] = cpu_baseline(cfg, p, args.nmf_iter, args.iters, args.cpu_sample)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))

    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
