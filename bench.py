"""
bench.py -- DegNorm NMF-OA hot path on MI355X.

    python bench.py [--config c2|c4] [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

--config c2 (default; BASELINE.json's metric, configs[1] / configs[2] when sharded): 20 000 synthetic genes x 10
samples, L ~ U[200, 5000], 5 DegNorm iterations, nmf_iter = 100.
--config c4 (configs[3]): 50 000 genes x 50 samples, L ~ U[501, 5000], take-every 500, 5 iterations.

One "step" = one complete DegNorm run over the resident genes: the ratio-SVD initialisation pass plus `--iters` (5)
outer iterations, each = the baseline-selection kernels over every gene of the rank's shard, the D2H of the DI rows /
flags / traces and the per-sample all-reduce (RCCL over xGMI under torchrun, also at N = 1).  Coverage is generated and
uploaded to HBM before the timed region (estimates are not fetched).  The genes are sharded across the N ranks by
length (utils.partition_by_length), so scaling is strong; `value` = total genes / max-over-ranks wall time.

Rank 0 prints ONE JSON line with the driver contract fields plus
  roofline      the dominant kernel against the resource that binds it (c2: fp64 vector issue, with the measured
                one-wave issue ceiling as a second peak and SURVEY 8(d)'s algorithmic bytes/s as a named secondary
                figure; c4: HBM bytes of the initial pass), `traffic` from the committed rocprofv3 PMC passes of THIS
                source tree (refused when the sources differ from the profiled ones),
  parity        after the clock stops: the oracle re-runs a sample of genes with the scale factors each timed outer
                iteration actually used, and the device's DI rows / branch traces of the LAST timed step are compared,
  cpu_baseline  the oracle on this box's host cores on a bounded sample.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (AMD MI355X spec)
# tools/ubench/clock_issue.hip on the box (profiles/round2/ubench_clock_issue.txt), in-kernel clock 2.37-2.39 GHz: ONE wave
# alone on a SIMD issues the instruction mix of a column of the pass (10 cvt + 10 mul + 30 fma + 10 max + 55 Gram fma,
# registers only) at 4.22 cycles per instruction against the 4 of the peak above (two waves per SIMD: 3.95); a stream of
# nothing but dependent-free fp64 FMAs issues slower (5.75, 4.4 with four waves).  The kernel runs one wave per SIMD.
FP64_ISSUE_CYCLES_1WAVE = 4.22
ROUND = 'round2'


def source_hash():
    """sha256 over the kernel sources and the build recipe: identifies the binary build() makes from this tree."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'degnorm_amd', 'csrc', '*.h*'))) + [os.path.join(ROOT, 'degnorm_amd', 'build.py')]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 22), b''):
            h.update(blk)
    return h.hexdigest()


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


def pmc_traffic(config, kernel_name, genes_in_kernel):
    """
    Fabric-side bytes per launch of the dominant kernel from the PMC counters.  bench.py cannot collect PMCs itself: the
    figure comes from committed rocprofv3 passes of THIS command (tools/profile_round.sh -> profiles/<round>/
    pmc_traffic_<config>.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950
    calibration).  It is reported only when the profile was taken from the same kernel sources (source_hash), kernel
    and gene count as the run that just finished; otherwise `traffic` is null and the reason is given.
    FETCH_SIZE / WRITE_SIZE are fabric-side counters: Infinity-Cache hits are included (MI355X guide, HBM section).
    """
    path = os.path.join(ROOT, 'profiles', ROUND, 'pmc_traffic_{0}.json'.format(config))
    info = {'profile': os.path.relpath(path, ROOT), 'side': 'fabric (L2 <-> Infinity Fabric; includes Infinity-Cache hits)'}
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        info['refused'] = 'no committed PMC profile for this configuration'
        return None, info
    src = source_hash()
    from degnorm_amd import _lib
    info['source_sha256'] = src
    info['profiled_source_sha256'] = d.get('source_sha256')
    info['lib_sha256_match'] = d.get('lib_sha256') == file_sha256(_lib.LIB_PATH)
    if d.get('source_sha256') != src:
        info['refused'] = 'kernel sources changed since the profile was taken'
        return None, info
    k = (d.get('kernels') or {}).get(kernel_name)
    if k is None or int(k.get('genes_in_kernel') or -1) != int(genes_in_kernel):
        info['refused'] = 'profiled kernel / gene count differ from this run'
        return None, info
    info['read_bytes'] = k.get('read_bytes_per_launch')
    info['write_bytes'] = k.get('write_bytes_per_launch')
    return float(k['hbm_bytes_per_launch']), info


def host_cores():
    from oracle import oracle as orc
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
    return cores


def cpu_baseline(cfg, name, p, nmf_iter, iters, rate, n_sample):
    """The CPU oracle (oracle/, parity-pinned port of the reference) timed on this box's host cores."""
    from oracle import oracle as orc
    from degnorm_amd import synth
    orc.build()
    cores = host_cores()
    covs = [synth.synth_gene(cfg['seed'], g, p, cfg['l_min'], cfg['l_max'])[0] for g in range(n_sample)]
    reads = np.vstack([synth.read_counts_from_coverage(c) for c in covs])
    ds = None
    if rate > 1:
        ds = np.random.RandomState(123).randint(0, rate, size=(iters, n_sample)).astype(np.int64)
    t0 = time.time()
    orc.run(covs, reads, degnorm_iter=iters, nmf_iter=nmf_iter, downsample_rate=rate, min_high_coverage=2 if rate > 1 else 50,
            ds_starts=ds, n_threads=cores)
    dt = time.time() - t0
    return {'value': n_sample / dt, 'unit': 'genes/sec', 'cores': cores, 'kind': 'port',
            'sample': 'first {0} genes of the {1} generator, full run ({2} outer iterations, nmf_iter {3}{4}), '
                      'oracle/nmfoa_oracle.c with OpenMP over genes, {5:.1f} s wall'
                      .format(n_sample, name, iters, nmf_iter, ', take-every {0}'.format(rate) if rate > 1 else '', dt)}


def parity_sample(lengths, k):
    """Genes at evenly spaced length quantiles: both gene classes and the whole work queue."""
    n = len(lengths)
    by_len = np.argsort(lengths, kind='stable')
    return np.unique(by_len[np.linspace(0, n - 1, min(k, n)).round().astype(int)])


def parity_check(eng, pick, cfg, p, my_genes, lengths, split, args, rate):
    """
    Post-clock value check of the LAST timed step.  For every outer iteration of that step the oracle processes a sample
    of this rank's genes (chosen before the clock started; the engine kept their raw device rows) with the scale factors
    (and down-sampling offsets) the device used in that iteration; compared: the raw DI rows (relative, BASELINE tolerance
    1e-5), ran_baseline_selection flags and the branch trace {n_hi_cov, #nmf calls, sum of active columns, exit code,
    loop-exit reason, #drops, status} exactly.
    """
    from oracle import oracle as orc
    from degnorm_amd import synth
    orc.build()
    covs = [synth.synth_gene(cfg['seed'], my_genes[j], p, cfg['l_min'], cfg['l_max'])[0] for j in pick]
    prm = orc.make_params(args.nmf_iter, 20, 2 if rate > 1 else 50, rate, False)
    cores = host_cores()
    max_rel, max_abs, flips, checked = 0.0, 0.0, 0, 0
    t0 = time.time()
    for i in range(len(eng.scale_hist)):
        ds = eng.offsets_hist[i][pick] if (rate > 1 and eng.offsets_hist[i] is not None) else None
        rho_o, flags_o, trace_o, _ = orc.baseline_batch(covs, eng.scale_hist[i], prm, ds_start=ds, n_threads=cores)
        rho_d, flags_d, trace_d = eng.rho_raw_hist[i], eng.flags_hist[i], eng.traces[i][pick]
        d = np.abs(rho_d - rho_o)
        max_abs = max(max_abs, float(d.max()))
        max_rel = max(max_rel, float((d / np.maximum(np.abs(rho_o), 1e-6)).max()))
        flips += int(np.sum(np.any(trace_d[:, :7] != trace_o[:, :7], axis=1) | (flags_d != flags_o)))
        checked += len(pick)
    wide = int(np.sum(lengths[pick] > split)) if split > 0 else len(pick)
    tiny = eng.dev.tiny_length() if (split > 0 and hasattr(eng.dev, 'tiny_length')) else 0
    n_pair = int(np.sum(lengths[pick] <= tiny)) if tiny > 0 else 0
    return {'max_rel_di': max_rel, 'max_abs_di': max_abs, 'branch_flips': flips, 'genes_checked': len(pick),
            'gene_iterations_checked': checked, 'outer_iterations': len(eng.scale_hist), 'genes_in_wide_class': wide, 'genes_in_pair_class': n_pair,
            'tolerance_rel_di': 1e-5, 'ok': bool(flips == 0 and max_rel < 1e-5),
            'what': 'last timed step; oracle (kind: port, pinned to the reference goldens) given each iteration\'s device-side '
                    'scale factors; DI rows unclipped; trace[:7] and flags exact', 'oracle_s': time.time() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2', choices=['c2', 'c4'])
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=2)    # the first step after the first-ever one still carries ~18 ms of one-time host cost
    ap.add_argument('--genes', type=int, default=0, help='total genes (default: the configuration\'s: 20000 / 50000)')
    ap.add_argument('--iters', type=int, default=5, help='outer DegNorm iterations per step')
    ap.add_argument('--nmf-iter', type=int, default=100)
    ap.add_argument('--cpu-sample', type=int, default=-1, help='genes in the CPU-baseline sample (0 = skip; default 768 / 2048)')
    ap.add_argument('--parity-genes', type=int, default=160, help='genes in the post-clock parity sample (0 = skip)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='collective backend under torchrun: nccl = RCCL (the measured path); gloo only to rehearse N > 1 on ONE GPU '
                         '(every rank on device 0; the line is then marked "rehearsal")')
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

    rehearsal = args.backend == 'gloo'
    if rehearsal:
        local_rank = 0                                                  # all ranks share the one GPU
    torch.cuda.set_device(local_rank)
    comm = LocalComm()
    distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ     # under torchrun use RCCL even at N = 1
    rccl = None
    if distributed:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group('gloo')
            comm = TorchComm(device='cpu')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            comm = TorchComm(device='cuda:{0}'.format(local_rank))
        rccl = {'backend': dist.get_backend(), 'rccl_ranks': dist.get_world_size(),
                'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()), 'rehearsal_on_one_gpu': rehearsal}

    cfg = dict(synth.CONFIGS[args.config])
    p = cfg['p']
    rate = 500 if args.config == 'c4' else 1
    n_genes = args.genes if args.genes > 0 else cfg['n_genes']
    # every rank gets the same total gene length (the generator's lengths are a cheap pure function of the gene id)
    all_len = [synth.gene_length(cfg['seed'], g, cfg['l_min'], cfg['l_max']) for g in range(n_genes)] if world > 1 else None
    parts = partition_by_length(all_len, world) if world > 1 else [list(range(n_genes))]
    my_genes = parts[rank] if rank < len(parts) else []
    t_gen = time.time()
    packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], my_genes, p, cfg['l_min'], cfg['l_max'],
                                                   n_threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    t_gen = time.time() - t_gen

    eng = ShardedNMFOA(comm=comm, device=local_rank, degnorm_iter=args.iters, nmf_iter=args.nmf_iter, downsample_rate=rate)
    t_up = time.time()
    eng.load_packed(packed, lengths, p, reads, global_ids=np.asarray(my_genes, dtype=np.int64), n_total=n_genes)
    t_up = time.time() - t_up
    del packed

    def sync():
        torch.cuda.synchronize()
        comm.Barrier()
        torch.cuda.synchronize()

    init_ms = []

    def step():
        eng.initialize()
        init_ms.append(eng.dev.last_init_ms())
        for i in range(args.iters):
            eng.iterate(i, want_estimates=False)

    for _ in range(args.warmup):
        step()

    split = eng.dev.split_length()
    tiny = eng.dev.tiny_length() if split > 0 else 0
    wide = lengths > split if split > 0 else np.ones(len(lengths), dtype=bool)
    pair = (lengths <= tiny) & ~wide                                    # class 2: one wavefront per gene, two genes per workgroup
    kernel_ms, narrow_ms, pair_ms, all_traces, eng_span = [], [], [], [], []
    pick = parity_sample(lengths, args.parity_genes) if (args.parity_genes > 0 and len(lengths) > 0) else None
    eng.history_rows = pick
    init_ms.clear()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        step()
        kernel_ms += [c[0] for c in eng.class_ms]
        narrow_ms += [c[1] for c in eng.class_ms]
        pair_ms += [c[2] for c in eng.class_ms]
        eng_span += list(eng.span_ms)
        all_traces += eng.traces                                       # accounting happens after the clock stops
    sync()
    dt = time.time() - t0

    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = n_genes * args.steps / dt
        avg_ms = float(np.mean(kernel_ms))
        name0 = eng.dev.class_kernel_name(0)
        out = {
            'metric': 'genes/sec (20k genes x 10 samples, 5 iters)' if args.config == 'c2'
                      else 'genes/sec (50k genes x 50 samples, downsample-grid 500, 5 iters)',
            'value': value, 'unit': 'genes/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config {0}: {1} synthetic genes x {2} samples, L~U[{3},{4}], {5} DegNorm iters, nmf_iter {6}{7}, '
                                   'fp32 coverage in HBM, fp64 arithmetic'
                                   .format(2 if args.config == 'c2' else 4, n_genes, p, cfg['l_min'], cfg['l_max'], args.iters,
                                           args.nmf_iter, ', take-every {0}'.format(rate) if rate > 1 else ''),
                       'genes_per_gpu': len(my_genes),
                       'sharding': 'length-balanced gene partition, 1 all-reduce of 3p+3 f64 per outer iter'},
            'rccl': rccl,
        }
        if args.config == 'c2':
            # Three gene classes = three kernels per outer iteration on three streams: class 0 (genes longer than the split
            # length, 256-thread workgroups, launched first), class 1 (128-thread workgroups, two per CU) and class 2 (the
            # shortest genes, one wavefront per gene, two genes per 128-thread workgroup).  Classes 1 and 2 are launched
            # right behind class 0 and take over the CUs as it drains, so the one that ends last spans the whole sweep.  The
            # roofline object describes that kernel (the longest launch) and puts ALL kernels' work over its duration.
            cls_mask = [wide, ~wide & ~pair, pair]
            cls_ms = [float(np.mean(kernel_ms)), float(np.mean(narrow_ms)), float(np.mean(pair_ms)) if pair_ms else 0.0]
            dom = int(np.argmax(cls_ms))
            others = [c for c in range(3) if c != dom and cls_mask[c].any()]
            mask, avg_ms, name_d = cls_mask[dom], cls_ms[dom], eng.dev.class_kernel_name(dom)
            span_ms = float(np.mean(eng_span)) if eng_span else max(cls_ms)

            def work(m):
                # fp64 vector work of the inner passes: per column and inner iteration u.a (2p), the update (5p), the Gram
                # update (p(p+1)) and the 1/s scaling (p); wave-instructions: p(p+1)/2 + 5p FMA/max/mul + p cvt per 64 columns
                col_iters = float(np.mean([float(tr[m, 2].astype(np.float64).sum()) * args.nmf_iter for tr in all_traces]))
                return col_iters * (p * p + 9.0 * p), col_iters / 64.0 * (p * (p + 1) / 2.0 + 6.0 * p)
            flop_d, instr_d = work(mask)
            flop_a, instr_a = work(np.ones(len(lengths), dtype=bool))
            alg_d = float(np.mean([algorithmic_bytes(tr, lengths, p, args.nmf_iter, mask) for tr in all_traces]))
            alg_all = float(np.mean([algorithmic_bytes(tr, lengths, p, args.nmf_iter) for tr in all_traces]))
            tflops = flop_a / (avg_ms * 1e-3) / 1e12
            simd_cycles = lambda ms: ms * 1e-3 * 2.4e9 * 256 * 4
            issue_peak = FP64_VECTOR_PEAK_TFLOPS * 4.0 / FP64_ISSUE_CYCLES_1WAVE
            traffic, tinfo = pmc_traffic(args.config, name_d, int(mask.sum())) if (world == 1 and n_genes == cfg['n_genes']) else (None, {'refused': 'not the profiled shard'})
            traffic_o = {eng.dev.class_kernel_name(c): (pmc_traffic(args.config, eng.dev.class_kernel_name(c), int(cls_mask[c].sum()))[0]
                                                        if traffic is not None else None) for c in others}
            traffic_pair = (traffic + sum(traffic_o.values())) if (traffic is not None and all(v is not None for v in traffic_o.values())) else None
            out['roofline'] = {
                'bound': 'fp64_valu', 'achieved': tflops, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tflops / FP64_VECTOR_PEAK_TFLOPS,
                'what': 'fp64 vector work of ALL genes (every class kernel) / average launch duration of the dominant kernel: the '
                        'class kernels are launched back to back on their own streams, the dominant one is the last to end, so its '
                        'launch spans the whole sweep and the other kernels run INSIDE that window',
                'kernel': name_d, 'avg_launch_ms': avg_ms, 'launches_timed': len(kernel_ms),
                'genes_in_kernel': int(mask.sum()), 'split_length': split, 'pair_length': tiny,
                'concurrent_kernels': [{'kernel': eng.dev.class_kernel_name(c), 'genes': int(cls_mask[c].sum()),
                                        'avg_launch_ms': cls_ms[c]} for c in others],
                'sweep_span_ms': span_ms,
                'dominant_kernel_own_work': {'fp64_tflops': flop_d / (avg_ms * 1e-3) / 1e12,
                                             'frac': flop_d / (avg_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                             'note': 'only the dominant kernel\'s genes over its launch duration (it has the chip to itself '
                                                     'for only part of that time)'},
                'issue_ceiling': {'peak': issue_peak, 'frac': tflops / issue_peak,
                                  'what': 'one wave per SIMD issues the column\'s instruction mix at {0} cycles per instruction, not 4 '
                                          '(tools/ubench/clock_issue.hip, in-kernel clock 2.37-2.39 GHz; profiles/{1}/ubench_clock_issue.txt)'
                                          .format(FP64_ISSUE_CYCLES_1WAVE, ROUND)},
                'valu_issue_slots': {'wave_instructions': instr_a, 'slots_frac': instr_a * 4.0 / simd_cycles(avg_ms),
                                     'what': 'fp64 wave-instructions of the passes x 4 cycles / (sweep time x 2.4 GHz x 1024 SIMDs)'},
                'flop_per_column_iteration': p * p + 9.0 * p,
                'traffic': traffic_pair, 'traffic_info': tinfo,
                'traffic_per_kernel': dict({name_d: traffic}, **traffic_o),
                'traffic_rate_gbps': (traffic_pair / (avg_ms * 1e-3) / 1e9) if traffic_pair else None,
                'traffic_frac_of_hbm_peak': (traffic_pair / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_pair else None,
                'hbm_algorithmic': {'bytes_per_sweep': alg_all, 'rate_gbps': alg_all / (avg_ms * 1e-3) / 1e9, 'hbm_peak_gbps': HBM_PEAK_GBPS,
                                    'ratio_to_hbm_peak': alg_all / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    'what': 'SURVEY 8(d) algorithmic bytes (fp32 x and lambda re-streamed every inner iteration) / sweep time; '
                                            'NOT a roofline fraction: the kernels keep x + lambda on chip (registers + LDS), so these bytes are '
                                            'mostly never moved'},
                'note': 'rank-0 shard; HIP events on the library streams around each launch'}
        else:
            # config 4: after the row-maxima shortcut an outer iteration reads only the sampled columns; the kernel that
            # streams HBM is the initial ratio-SVD pass over the whole transcripts (nmf.py:109-121, :522-525)
            init_avg = float(np.mean(init_ms))
            alg_init = 8.0 * p * float(lengths.sum())                           # init pass 8 p L_g per gene
            sampled = float(np.mean([tr[:, 0].astype(np.float64).sum() for tr in all_traces]))
            out['roofline'] = {
                'bound': 'hbm', 'achieved': alg_init / (init_avg * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': alg_init / (init_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'traffic': None,
                'kernel': eng.dev.init_kernel_name(), 'avg_launch_ms': init_avg, 'launches_timed': len(init_ms),
                'algorithmic_bytes_per_launch': alg_init,
                'iteration_kernel': {'kernel': name0, 'avg_launch_ms': avg_ms, 'launches_timed': len(kernel_ms),
                                     'active_columns_per_launch': sampled,
                                     'bytes_needed_per_launch': 4.0 * p * float(np.ceil(lengths / float(rate)).sum()),
                                     'note': 'row maxima once per upload + only the sampled columns are read (SURVEY 8(d) shortcut, declared): '
                                             'this kernel is bound by the latency of ~100 tiny dependent eigen-solves per nmf() call, not by HBM'},
                'shortcut': 'max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i): per-iteration full-length scans removed; without it SURVEY 8(d) '
                            'counts 8 p L_g per gene and outer iteration = {0:.3e} B per launch'.format(alg_init),
                'note': 'rank-0 shard; HIP events on the library stream'}
        out['setup'] = {'synth_s': t_gen, 'upload_s': t_up}
        out['parity'] = parity_check(eng, pick, cfg, p, my_genes, lengths, split, args, rate) if pick is not None else None
        n_cpu = args.cpu_sample if args.cpu_sample >= 0 else (768 if args.config == 'c2' else 2048)
        out['cpu_baseline'] = cpu_baseline(cfg, args.config, p, args.nmf_iter, args.iters, rate, n_cpu) if (world == 1 and n_cpu > 0) else None
        print(json.dumps(out))

    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
