"""
bench.py -- DegNorm NMF-OA hot path on MI355X.

    python bench.py [--config c2|c4] [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before touching torch or the GPU it
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
as a child process (one rank per GPU over RCCL), relays rank 0's single JSON line to its own stdout and exits non-zero if
any rank fails (the reference's counterpart: `mpiexec -n R degnorm_mpi`, __main_mpi__.py:29-43, nmf_mpi.py:555-629).
Started under torchrun (WORLD_SIZE set) it is one rank.

--config c2 (default; BASELINE.json's metric, configs[1] / configs[2] when sharded): 20 000 synthetic genes x 10
samples, L ~ U[200, 5000], 5 DegNorm iterations, nmf_iter = 100.
--config c4 (configs[3]): 50 000 genes x 50 samples, L ~ U[501, 5000], take-every 500, 5 iterations.

One "step" = one complete DegNorm run over the resident genes: the ratio-SVD initialisation pass, `--iters` (5) outer
iterations -- each = the baseline-selection kernels over every gene of the rank's shard, the device-side outer update and the
per-sample all-reduce (RCCL, also at N = 1) -- and the D2H of the final DI scores / adjusted counts / flags (fetch_state).
Coverage is generated and uploaded to HBM before the timed region; estimates are not fetched (reported in `end_to_end`).
The genes are sharded across the N ranks by predicted cost (utils.partition_by_cost), so scaling is strong; `value` =
total genes / max-over-ranks wall time.

Rank 0 prints ONE JSON line with the driver contract fields plus
  roofline      the dominant kernel against the resource that binds it (c2: fp64 vector issue, with the measured
                one-wave issue ceiling as a second peak and SURVEY 8(d)'s algorithmic bytes/s as a named secondary
                figure; c4: HBM bytes of the initial pass + the latency bound of the iteration kernel), `traffic` from the
                committed rocprofv3 PMC passes of THIS source tree (refused when the sources differ from the profiled ones),
  parity        after the clock stops: (i) the oracle re-runs a sample of genes with the scale factors each timed outer
                iteration actually used (kernels alone), (ii) `chain`: the whole run -- initial pass, 5 iterations, device-side
                outer update -- on the CPU-baseline sample against the oracle's own run, flipped genes counted and measured,
                (iii) `tie_sensitive`: genes with a column exactly on the 0.1 x max threshold of get_high_coverage_idx,
  end_to_end    GeneNMFOA.fit(cov_dict, reads) on the full float64 coverage dict: pack + H2D, run, estimates, D2H,
  cpu_baseline  the oracle on this box's host cores (all cores, and one thread) on a bounded sample,
  also          (default run only) config 4 measured in the same process after config 2.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (AMD MI355X spec)
# tools/ubench/instr_cost.hip on the box (profiles/round3/ubench_instr_cost.txt): ONE wave alone on a SIMD issues every vector
# instruction of the column body -- fp64 FMA / mul / max with any operand mix, conversions, the 32-bit AGPR moves and unpacks, even
# back-to-back dependent FMAs -- at 4.08 cycles, against the 4 of the peak above (the 4.22 of round 2's clock_issue.hip included the
# scalar overhead of its loop).  The kernel runs one wave per SIMD, so a moved dword costs as much as an fp64 FMA.
FP64_ISSUE_CYCLES_1WAVE = 4.08
ROUND = 'round4'
REFERENCE_PY_GENES_PER_S_PER_THREAD = 0.18    # the reference itself (n_jobs = 1) on config-2-like genes, 5 iterations


# ---------------------------------------------------------------------------------------------------------------------
# launcher (no torch, no GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, launcher=None):
    """
    Start the N rank processes as FRESH children (never exec: the GPU box forbids replacing a process) and relay rank 0's
    JSON line.  `launcher` replaces the `python -m torch.distributed.run ...` prefix (tests use a stub); the environment
    variable DN_BENCH_LAUNCHER (a JSON list) does the same for a parent started as a script.  Returns the exit code.
    """
    if launcher is None and os.environ.get('DN_BENCH_LAUNCHER'):
        launcher = json.loads(os.environ['DN_BENCH_LAUNCHER'])
    if launcher is None:
        launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
                    '--master-addr', '127.0.0.1', '--master-port', str(_free_port())]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')            # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(1, n))))
    env['DN_BENCH_PARENT'] = str(os.getpid())
    cmd = list(launcher) + [os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, universal_newlines=True, env=env)
    line = None
    for raw in proc.stdout:
        text = raw.strip()
        is_line = False
        if text.startswith('{'):
            try:
                is_line = 'metric' in json.loads(text)
            except ValueError:
                pass
        if is_line:
            line = text
        else:
            sys.stderr.write(raw)                                 # whatever else the ranks print goes to stderr
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write('bench.py: the rank launcher exited with code {0}\n'.format(rc))
        return rc if 0 < rc < 256 else 1
    if line is None:
        sys.stderr.write('bench.py: the ranks finished without printing a result line\n')
        return 3
    print(line)
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# accounting helpers
# ---------------------------------------------------------------------------------------------------------------------
def source_hash():
    """sha256 over the kernel sources (the templates, the reduction, the generated DPP products, the instantiation unit of the class
    kernels, the run-time-p family of config 4) and the build recipe with its flags: identifies the kernel code the PMC passes were
    taken on.  The host side (dn_api.hip) is not part of it; what the host side decides -- which genes run in which class kernel --
    is checked separately (kernel names and gene counts of the profile against the run)."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, 'degnorm_amd', 'csrc', f) for f in ('dn_inst.hip', 'dn_kernels.hpp', 'dn_reduce.hpp', 'dn_dpp_ops.hpp', 'dn_generic.hip')] + \
            [os.path.join(ROOT, 'degnorm_amd', 'build.py')]
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
    `mask` selects the genes one kernel processes (the gene classes run in separate launches).
    """
    sum_cols = trace[:, 2].astype(np.float64)
    per_gene = 4.0 * p * (2.0 * lengths + sum_cols * (3.0 * nmf_iter + 3.0))
    return float(per_gene[mask].sum() if mask is not None else per_gene.sum())


def pmc_traffic(config, kernel_name, genes_in_kernel):
    """
    Fabric-side bytes per launch of a kernel from the PMC counters.  bench.py cannot collect PMCs itself: the
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


_HOST_CORES = None


def host_cores():
    """Threads the oracle may use on this box; asked once (a later single-thread oracle run lowers OpenMP's own maximum)."""
    global _HOST_CORES
    if _HOST_CORES is None:
        _HOST_CORES = _host_cores()
    return _HOST_CORES


def _host_cores():
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


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + whole-chain parity (post-clock)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, name, p, nmf_iter, iters, rate, n_sample, n_single):
    """
    The CPU oracle (oracle/, parity-pinned port of the reference) timed on this box's host cores: all cores on `n_sample`
    genes and ONE thread on the first `n_single` of them (the reference-equivalent mode: the reference's own multiprocessing
    is thread-based and GIL-bound).  Returns (the cpu_baseline object, the oracle's run on the sample -- reused by
    the chain parity check --, the sample's coverage matrices and read counts).
    """
    from oracle import oracle as orc
    from degnorm_amd import synth
    orc.build()
    cores = host_cores()
    covs = [synth.synth_gene(cfg['seed'], g, p, cfg['l_min'], cfg['l_max'])[0] for g in range(n_sample)]
    reads = np.vstack([synth.read_counts_from_coverage(c) for c in covs])
    ds = None
    if rate > 1:
        ds = np.random.RandomState(123).randint(0, rate, size=(iters, n_sample)).astype(np.int64)
    kw = dict(degnorm_iter=iters, nmf_iter=nmf_iter, downsample_rate=rate, min_high_coverage=2 if rate > 1 else 50)
    hist = {}
    t0 = time.time()
    ref = orc.run(covs, reads, ds_starts=ds, n_threads=cores, history=hist, **kw)
    dt = time.time() - t0
    ref['history'] = hist
    ref['ds_starts'] = ds
    single = None
    if n_single > 0:
        k = min(n_single, n_sample)
        t1 = time.time()
        orc.run(covs[:k], reads[:k], ds_starts=None if ds is None else ds[:, :k], n_threads=1, **kw)
        d1 = time.time() - t1
        single = {'value': k / d1, 'unit': 'genes/sec', 'cores': 1, 'sample': 'first {0} genes, {1:.1f} s wall'.format(k, d1)}
    tail = ', take-every {0}'.format(rate) if rate > 1 else ''
    out = {'value': n_sample / dt, 'unit': 'genes/sec', 'cores': cores, 'kind': 'port',
           'sample': 'first {0} genes of the {1} generator, full run ({2} outer iterations, nmf_iter {3}{4}), '
                     'oracle/nmfoa_oracle.c with OpenMP over genes, {5:.1f} s wall.  Beside it: `single_thread` = the same port on '
                     'one thread; the reference\'s own Python (scipy ARPACK, n_jobs = 1) measured in the build container on '
                     'config-2-like genes = {6} genes/s per thread (BASELINE.md section 2, not re-run here)'
                     .format(n_sample, name, iters, nmf_iter, tail, dt, REFERENCE_PY_GENES_PER_S_PER_THREAD),
           'single_thread': single,
           'reference_python_genes_per_s_per_thread': REFERENCE_PY_GENES_PER_S_PER_THREAD if name == 'c2' else None}
    return out, ref, covs, reads


def chain_parity(ref, covs, reads, nmf_iter, iters, rate, device):
    """
    The whole chain on the device against the oracle's own run of the same genes: initial ratio-SVD pass and normalisation,
    every outer iteration's kernels, dn_outer_partials -> all-reduce -> dn_outer_apply, fetch_state -- through
    GeneNMFOA.fit().  Unlike `parity_check` nothing is handed over between the two runs: each follows its OWN scale factors.
    A gene is `flipped` when its branch trace differs in any iteration (a 0.1 x max threshold tie decided by the last bit of a
    scale factor); flipped genes are counted and their DI distance reported, NOT excluded silently.
    """
    from collections import OrderedDict
    from degnorm_amd.nmf import GeneNMFOA
    n = len(covs)
    m = GeneNMFOA(degnorm_iter=iters, nmf_iter=nmf_iter, downsample_rate=rate, device=device)
    if ref.get('ds_starts') is not None:
        m.downsample_offsets = ref['ds_starts']
    t0 = time.time()
    m.fit(OrderedDict(('g%06d' % k, c) for k, c in enumerate(covs)), reads)
    dt = time.time() - t0
    flipped = np.zeros(n, dtype=bool)
    for i in range(iters):
        flipped |= np.any(m.traces[i][:, :7] != ref['history']['trace'][i][:, :7], axis=1)
    flipped |= np.any(m.ran_baseline_selection != ref['ran_baseline_selection'], axis=1)
    detail = []
    for k in np.flatnonzero(flipped)[:4]:                               # what differs, and from which iteration on
        for i in range(iters):
            td, to = m.traces[i][k, :7], ref['history']['trace'][i][k, :7]
            if np.any(td != to) or m.ran_baseline_selection[k, i] != ref['ran_baseline_selection'][k, i]:
                detail.append({'gene': int(k), 'first_iteration': i + 1, 'device_trace': td.tolist(), 'oracle_trace': to.tolist(),
                               'fields': 'n_hi_cov, nmf calls, sum of active columns, exit code, loop-exit reason, drops, status'})
                break
    ok = ~flipped
    rel = np.abs(m.rho - ref['rho']) / np.maximum(np.abs(ref['rho']), 1e-6)
    rel_adj = np.abs(m.x_adj - ref['x_adj']) / np.maximum(np.abs(ref['x_adj']), 1e-300)
    out = {'genes': n, 'outer_iterations': iters,
           'max_rel_di_final': float(rel[ok].max()) if ok.any() else None,
           'max_rel_adjusted_counts': float(rel_adj[ok].max()) if ok.any() else None,
           'max_rel_scale_factors': float(np.max(np.abs(m.scale_factors - ref['scale_factors']) / ref['scale_factors'])),
           'flipped_genes': int(flipped.sum()),
           'flipped_max_abs_di': float(np.abs(m.rho - ref['rho'])[flipped].max()) if flipped.any() else 0.0,
           'flipped_max_rel_adjusted_counts': float(rel_adj[flipped].max()) if flipped.any() else 0.0,
           'flipped_detail': detail, 'device_s': dt,
           'what': 'GeneNMFOA.fit() on the cpu_baseline sample vs the oracle\'s own full run (each side follows its own scale '
                   'factors through all outer iterations, device-side outer update included); flipped = branch trace or flags differ'}
    out['ok'] = bool(out['max_rel_scale_factors'] < 1e-5 and (out['max_rel_di_final'] or 0.0) < 1e-5)
    return out


def tie_sensitive_genes(packed, lengths, p, scale):
    """
    Genes with a column sitting exactly on the threshold of get_high_coverage_idx (nmf.py:66-76: max_i F_ij > 0.1 max F,
    F = x / s): the comparison of such a column is decided by the last bit of the quotients, so ANY change of the scale
    factors in the 16th digit (another BLAS, another summation order) may move the gene onto another branch -- in the
    reference as well.  Counted at the given scale factors, within 4 ulp of the threshold.
    """
    inv = (1.0 / np.asarray(scale, dtype=np.float64))[:, None]
    n_sens, n_cols, o = 0, 0, 0
    for L in lengths:
        L = int(L)
        F = packed[o:o + p * L].reshape(p, L).astype(np.float64) * inv
        o += p * L
        cm = F.max(axis=0)
        thr = 0.1 * cm.max()
        near = np.abs(cm - thr) <= 8.9e-16 * thr
        k = int(near.sum())
        n_cols += k
        n_sens += k > 0
    return {'genes': int(n_sens), 'columns': int(n_cols), 'of_genes': int(len(lengths)),
            'what': 'genes with >= 1 column whose scaled maximum is within 4 ulp of 0.1 x max F at the final scale factors: '
                    'their branch is decided by round-off of the scale factors (nmf.py:76), in the reference too'}


def pileup_parity(device, p, scale, nmf_iter, n):
    """
    Read pile-up coverage (synth.pileup_gene: reads of 75-150 bases stacked into piecewise-constant small integers -- DegNorm's
    real input kind, pinned with the reference in tests/golden/pileup.npz) through one outer iteration on the device and on the
    oracle with the run's final scale factors: branch flips, DI distance, and how many of these genes are tie-sensitive
    (a column within 4 ulp of the 0.1 x max F threshold) -- on small-integer coverage 10 x == max ties are the rule.
    """
    from oracle import oracle as orc
    from degnorm_amd import synth, _lib
    covs = [synth.pileup_gene(13, g, p, 300, 3000)[0] for g in range(n)]
    dev = _lib.Device(device)
    try:
        dev.upload(covs)
        rho_d, flags_d, tr_d = dev.baseline_iteration(scale, nmf_iter=nmf_iter)
    finally:
        dev.close()
    t0 = time.time()
    rho_o, flags_o, tr_o, _ = orc.baseline_batch(covs, scale, orc.make_params(nmf_iter, 20, 50, 1, False), n_threads=host_cores())
    flips = np.any(tr_d[:, :7] != tr_o[:, :7], axis=1) | (flags_d != flags_o)
    ok = ~flips
    d = np.abs(rho_d - rho_o)
    packed = np.concatenate([c.astype(np.float32).ravel() for c in covs])
    ties = tie_sensitive_genes(packed, np.array([c.shape[1] for c in covs]), p, scale)
    return {'genes': n, 'branch_flips': int(flips.sum()), 'max_rel_di': float((d[ok] / np.maximum(np.abs(rho_o[ok]), 1e-6)).max()) if ok.any() else None,
            'flipped_max_abs_di': float(d[flips].max()) if flips.any() else 0.0,
            'genes_through_the_drop_loop': int((tr_o[:, 1] > 1).sum()), 'tie_sensitive_genes': ties['genes'], 'tie_sensitive_columns': ties['columns'],
            'oracle_s': time.time() - t0,
            'what': 'synth.pileup_gene(13, g, p) g < n: one outer iteration at the run\'s final scale factors, device vs oracle; the same '
                    'generator is pinned with the reference (three stable runs per gene) in tests/golden/pileup.npz'}


def power_steps_stats(traces, nmf_iter):
    """
    What the eigen-solves of the timed steps needed: per gene, trace[7] = power steps summed over its solves, trace[1] = nmf() calls,
    each of nmf_iter + 1 solves (nmf.py:88-101) -- the mean over all solves and the distribution of the per-gene means (a cold solve of
    ~10-25 plain steps from the uniform vector is in every call's average: 1 of 101 solves).  The squaring solver of rounds 1-3 spent
    10 step equivalents per solve whatever the matrix.
    """
    steps = np.concatenate([tr[:, 7].astype(np.float64) for tr in traces])
    solves = np.concatenate([tr[:, 1].astype(np.float64) * (nmf_iter + 1) for tr in traces])
    m = solves > 0
    per_gene = steps[m] / solves[m]
    edges = [0, 3, 4, 5, 6, 7, 8, 10, 15, 1e9]
    hist = np.histogram(per_gene, bins=edges)[0]
    return {'mean': float(steps[m].sum() / solves[m].sum()), 'per_gene_mean_quantiles_10_50_90_99': [float(q) for q in np.percentile(per_gene, [10, 50, 90, 99])],
            'per_gene_mean_histogram': {'edges': edges[:-1] + ['inf'], 'gene_iterations': hist.tolist()},
            'what': 'plain power steps of top_eig_dpp per solve (its return value counts the steps plus one), warm start + shift'}


def parity_sample(lengths, k):
    """Genes at evenly spaced length quantiles: every gene class and the whole work queue."""
    n = len(lengths)
    by_len = np.argsort(lengths, kind='stable')
    return np.unique(by_len[np.linspace(0, n - 1, min(k, n)).round().astype(int)])


def parity_check(eng, pick, cfg, p, my_genes, lengths, split, nmf_iter, rate):
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
    prm = orc.make_params(nmf_iter, 20, 2 if rate > 1 else 50, rate, False)
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


def end_to_end(packed, lengths, p, reads, iters, nmf_iter, rate, device, reps=2):
    """
    SURVEY 8(d)(i): the call the degnorm CLI makes (__main__.py:264-270) -- GeneNMFOA.fit(cov_dict, reads) on the float64
    coverage dict of the WHOLE configuration: host packing (threads) + H2D, the run, the estimates of the last iteration
    (rebuild + D2H + list of views) and the D2H of rho / x_adj / x_weighted / flags.  `genes_per_s` excludes the
    estimates (as SURVEY defines the metric), `genes_per_s_with_estimates` includes them.
    """
    from collections import OrderedDict
    from degnorm_amd.nmf import GeneNMFOA
    t0 = time.time()
    cov_dat, o = OrderedDict(), 0
    for g, L in enumerate(lengths):
        L = int(L)
        cov_dat['gene_%06d' % g] = packed[o:o + p * L].reshape(p, L).astype(np.float64)      # what the CLI hands over (reads.py:714)
        o += p * L
    t_dict = time.time() - t0
    runs = []
    for _ in range(reps):
        m = GeneNMFOA(degnorm_iter=iters, nmf_iter=nmf_iter, downsample_rate=rate, device=device)
        t1 = time.time()
        est = m.fit(cov_dat, reads)
        total = time.time() - t1
        tm = dict(m.timings)
        tm['total_s'] = total
        tm['other_host_s'] = total - sum(m.timings.values())
        runs.append(tm)
        n_est = len(est)
        del est
        m._dev.close()
    best = min(runs, key=lambda r: r['total_s'])
    n = len(lengths)
    return {'pack_s': best['pack_upload_s'], 'upload_s': None, 'pack_upload_s': best['pack_upload_s'], 'run_s': best['run_s'],
            'fetch_state_s': best['fetch_state_s'], 'estimates_s': best['estimates_s'], 'other_host_s': best['other_host_s'],
            'total_s': best['total_s'],
            'genes_per_s': n / (best['total_s'] - best['estimates_s']), 'genes_per_s_with_estimates': n / best['total_s'],
            'estimates_returned': n_est, 'input_bytes_float64': int(8 * p * int(np.sum(lengths))), 'calls': runs,
            'dict_build_s_untimed': t_dict,
            'what': 'GeneNMFOA.fit(OrderedDict of float64 p x L matrices, reads) on the whole configuration, best of {0} calls (all '
                    'listed under `calls`; the first one also pays one-time allocations); pack_upload_s = float64 -> float32 packing '
                    'on the host threads into a pinned staging buffer + H2D + row maxima (one C call, dn_upload_ragged; `pack_s` '
                    'repeats it, `upload_s` is not separable)'.format(reps)}


def end_to_end_sharded(ctx, cfg, p, n_genes, args, rate):
    """
    The sharded API end to end (--sharded-api, N > 1): run_gene_nmfoa_mpi(comm, cov_dict, reads) as degnorm_mpi calls it
    (__main_mpi__.py:429-436) -- rank 0 holds the float64 coverage dict of the WHOLE configuration, ships every rank its packed
    share (raw float32 buffers, point to point), every rank runs its resident shard, rank 0 collects estimates / DI / adjusted
    counts / flags (raw buffers, to rank 0 only).  Per rank: the stage times, the bytes that rank sent and received, and its peak
    resident set -- no rank but 0 ever holds more than its own share.  Collective: every rank calls it.
    """
    import resource
    from collections import OrderedDict
    from degnorm_amd import synth
    from degnorm_amd.nmf_mpi import run_gene_nmfoa_mpi
    comm, rank, world = ctx['comm'], ctx['rank'], ctx['world']
    cov_dat, reads, in_bytes = None, None, 0
    if rank == 0:
        packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n_genes), p, cfg['l_min'], cfg['l_max'],
                                                       n_threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
        cov_dat, o = OrderedDict(), 0
        for g, L in enumerate(lengths):
            L = int(L)
            cov_dat['gene_%06d' % g] = packed[o:o + p * L].reshape(p, L).astype(np.float64)      # what the CLI hands over
            o += p * L
        in_bytes = int(8 * p * int(np.sum(lengths)))
        del packed
    sent0, recv0 = getattr(comm, 'bytes_sent', 0), getattr(comm, 'bytes_received', 0)
    tm = {}
    comm.Barrier()
    t0 = time.time()
    res = run_gene_nmfoa_mpi(comm, cov_dat, reads, degnorm_iter=args.iters, nmf_iter=args.nmf_iter, downsample_rate=rate,
                             device=ctx['local_rank'], timings=tm)
    total = time.time() - t0
    mine = dict(tm, rank=rank, total_s=total, bytes_sent=int(getattr(comm, 'bytes_sent', 0) - sent0),
                bytes_received=int(getattr(comm, 'bytes_received', 0) - recv0),
                peak_rss_mb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0)
    per_rank = comm.gather_objects(mine)
    if rank != 0:
        return None
    n_est = len(res['estimates'])
    del res, cov_dat
    return {'total_s': max(r['total_s'] for r in per_rank), 'genes_per_s_with_estimates': n_genes / max(r['total_s'] for r in per_rank),
            'input_bytes_float64': in_bytes, 'estimates_returned': n_est, 'per_rank': per_rank,
            'what': 'run_gene_nmfoa_mpi(comm, OrderedDict of float64 p x L matrices on rank 0, reads): scatter_s = checks + partition + '
                    'float32 packing + shipping the shares (rank 0) / waiting for and receiving the own share (workers); upload_s = H2D of '
                    'the share; run_s = initial pass + outer iterations + estimates of the last iteration + D2H; gather_s = results to rank 0 '
                    '(estimates float64, DI, adjusted counts, flags), raw buffers point to point; peak_rss_mb includes what the rank held '
                    'before the call (the bench shard, on rank 0 the float64 input dict)'}


# ---------------------------------------------------------------------------------------------------------------------
# one configuration on this rank
# ---------------------------------------------------------------------------------------------------------------------
def measure(config, args, ctx, steps, warmup, n_parity, n_cpu, n_single, want_e2e):
    """Generate, upload, time `steps` steps; rank 0 returns the result object (other ranks None)."""
    import torch
    from degnorm_amd import synth
    from degnorm_amd.nmf_mpi import ShardedNMFOA
    from degnorm_amd.utils import partition_by_cost

    rank, world, local_rank, comm, dist = ctx['rank'], ctx['world'], ctx['local_rank'], ctx['comm'], ctx['dist']
    cfg = dict(synth.CONFIGS[config])
    p = cfg['p']
    rate = 500 if config == 'c4' else 1
    n_genes = args.genes if args.genes > 0 else cfg['n_genes']
    # every rank gets the same predicted cost (the generator's lengths are a cheap pure function of the gene id)
    if world > 1:
        all_len = [synth.gene_length(cfg['seed'], g, cfg['l_min'], cfg['l_max']) for g in range(n_genes)]
        parts = partition_by_cost(all_len, world, p=p, downsample_rate=rate)
    else:
        parts = [list(range(n_genes))]
    my_genes = parts[rank] if rank < len(parts) else []
    t_gen = time.time()
    packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], my_genes, p, cfg['l_min'], cfg['l_max'],
                                                   n_threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    t_gen = time.time() - t_gen

    eng = ShardedNMFOA(comm=comm, device=local_rank, degnorm_iter=args.iters, nmf_iter=args.nmf_iter, downsample_rate=rate)
    eng.reuse_buffers = True        # every step fills the same host arrays (per-gene counters, final state): a step is a repeated run
    eng.trace_columns = 8           # per iteration the 8 counters of a gene come back (n_hi_cov, calls, columns, exit, ..., solver steps), not
                                    # the dropped-bin sequence behind them (diagnostics: 9.6 MB per iteration on config 4)
    eng.keep_packed = bool(args.redeal) and world > 1
    t_up = time.time()
    eng.load_packed(packed, lengths, p, reads, global_ids=np.asarray(my_genes, dtype=np.int64), n_total=n_genes)
    t_up = time.time() - t_up
    rowmax_ms = eng.dev.last_rowmax_ms() if eng.n_local > 0 else 0.0
    lib_comm = None
    if ctx['rccl_ranks'] is not None and not ctx['rehearsal'] and not args.torch_collective:
        # the collective of the step runs INSIDE the library (dn_comm_*: ncclAllReduce on the library's own stream, in place on its
        # device buffer); torch.distributed only carried the 128-byte communicator id to the ranks and times the barrier around the clock
        try:
            eng.attach_library_comm()
            lib_comm = eng.dev.comm_library()
        except Exception as e:                                          # no librccl the library can bind: the torch path of round 3 carries the step
            sys.stderr.write('bench.py: library-side collective unavailable ({0!r}); using torch.distributed on the device buffer\n'.format(e))
            eng._lib_comm = False
        ok = np.array([1.0 if lib_comm is not None else 0.0])
        if world > 1:                                                   # all ranks or none: a mixed state would deadlock the first reduction
            ok = comm.allreduce_sum(ok)
            if ok[0] < world and lib_comm is not None:
                eng.dev.comm_destroy()
                eng._lib_comm, lib_comm = False, None
    keep_packed = rank == 0 and world == 1 and config == 'c2'              # the float64 dict of end_to_end / the tie count are made from it
    if not keep_packed:
        packed = None

    def sync():
        torch.cuda.synchronize()
        comm.Barrier()
        torch.cuda.synchronize()

    init_ms, fetch_s = [], []

    def step():
        eng.initialize()
        init_ms.append(eng.dev.last_init_ms() if eng.n_local > 0 else 0.0)
        for i in range(args.iters):
            eng.iterate(i, want_estimates=False)
        tf = time.perf_counter()
        eng.fetch_state()                                               # final rho / x_adj / x_weighted / flags: D2H inside the clock
        fetch_s.append(time.perf_counter() - tf)

    redeal_info = None
    if args.redeal and world > 1:
        # --redeal: ONE untimed run hands over the few genes that level the measured per-gene cost of its first outer iteration
        # (ShardedNMFOA.redeal); the timed steps then run on that partition (a production run does this once, after iteration 1)
        eng.initialize()
        eng.iterate(0, want_estimates=False)
        redeal_info = eng.redeal()
        for i in range(1, args.iters):
            eng.iterate(i, want_estimates=False)
        eng.fetch_state()
        lengths, my_genes = eng._lengths.copy(), eng.global_ids.tolist()
    for _ in range(warmup):
        step()

    split = eng.dev.split_length() if eng.n_local > 0 else 0
    tiny = eng.dev.tiny_length() if split > 0 else 0
    # the measured stream-read ceiling of this device, beside the data sheet's 8 TB/s (a read-only kernel with four
    # 16-byte loads per lane in flight over 1 GiB, best of 5; outside the clock)
    stream_gbps = eng.dev.measure_read_gbps(1 << 30, 5) if (rank == 0 and eng.n_local > 0) else None
    wide = lengths > split if split > 0 else np.ones(len(lengths), dtype=bool)
    pair = (lengths <= tiny) & ~wide                                    # class 2: one wavefront per gene, two genes per workgroup
    kernel_ms, narrow_ms, pair_ms, all_traces, eng_span = [], [], [], [], []
    pick = parity_sample(lengths, n_parity) if (n_parity > 0 and len(lengths) > 0) else None
    eng.history_rows = pick
    init_ms.clear()
    fetch_s.clear()
    sync()
    t0 = time.time()
    for _ in range(steps):
        step()
        kernel_ms += [c[0] for c in eng.class_ms]
        narrow_ms += [c[1] for c in eng.class_ms]
        pair_ms += [c[2] for c in eng.class_ms]
        eng_span += list(eng.span_ms)
        all_traces += [tr[:, :8].copy() for tr in eng.traces]          # the counters the accounting below reads (it happens after the clock stops)
    sync()
    dt = time.time() - t0

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cpu' if ctx['rehearsal'] else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        counts = [None] * world
        dist.all_gather_object(counts, (len(my_genes), int(wide.sum()), int((~wide & ~pair).sum()), int(pair.sum()), float(np.sum(lengths))))
    else:
        counts = [(len(my_genes), int(wide.sum()), int((~wide & ~pair).sum()), int(pair.sum()), float(np.sum(lengths)))]
    if rank != 0:
        try:
            eng.dev.close()                                             # the handle (and its RCCL communicator) goes now, not at interpreter exit
        except Exception:
            pass
        if world > 1 and args.sharded_api:
            end_to_end_sharded(ctx, cfg, p, n_genes, args, rate)        # collective: rank 0 joins after its post-clock checks
        return None

    value = n_genes * steps / dt
    name0 = eng.dev.class_kernel_name(0)
    out = {
        'metric': 'genes/sec (20k genes x 10 samples, 5 iters)' if config == 'c2'
                  else 'genes/sec (50k genes x 50 samples, downsample-grid 500, 5 iters)',
        'value': value, 'unit': 'genes/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'config {0}: {1} synthetic genes x {2} samples, L~U[{3},{4}], {5} DegNorm iters, nmf_iter {6}{7}, '
                               'fp32 coverage in HBM, fp64 arithmetic'
                               .format(2 if config == 'c2' else 4, n_genes, p, cfg['l_min'], cfg['l_max'], args.iters,
                                       args.nmf_iter, ', take-every {0}'.format(rate) if rate > 1 else ''),
                   'genes_per_gpu': len(my_genes),
                   'sharding': 'cost-balanced gene partition (utils.partition_by_cost), 1 all-reduce of 3p+4 f64 per outer iter',
                   'redeal': redeal_info,
                   'per_rank': [{'genes': c[0], 'wide': c[1], 'narrow': c[2], 'pair': c[3], 'total_length': c[4]} for c in counts],
                   'step': 'initial pass + {0} outer iterations + D2H of the final rho / x_adj / x_weighted / flags '
                           '(fetch_state {1:.2f} ms per step)'.format(args.iters, 1e3 * float(np.mean(fetch_s)))},
        'rccl_ranks': ctx['rccl_ranks'],
        'rccl': dict(ctx['rccl'] or {}, all_reduce_on_device_buffer=(getattr(comm, 'device_reductions', 0) > 0 or eng.library_reductions > 0),
                     collective_inside_library=lib_comm is not None, library_rccl=lib_comm,
                     library_reductions_per_step=(eng.library_reductions // max(1, steps + warmup)) if lib_comm else 0) if ctx['rccl'] else None,
    }
    if config == 'c2':
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

        # fp64 vector work of the inner passes, per column and inner iteration: u.a (2p), the update (5p), the Gram update
        # (p(p+1)) -- and the 1/s scaling (p) only where the pass still has it: the register-tier cohorts (p <= 12) keep the T
        # loop's state in raw count units (csrc/dn_kernels.hpp DN_RAW_MAX_P).  Wave-instructions per 64 columns: p(p+1)/2 + 4p
        # FMA / max (+ p mul when scaled) + p cvt.
        raw_units = p <= 12
        flop_col = p * p + (8.0 if raw_units else 9.0) * p
        instr_col = p * (p + 1) / 2.0 + (5.0 if raw_units else 6.0) * p

        def work(m):
            col_iters = float(np.mean([float(tr[m, 2].astype(np.float64).sum()) * args.nmf_iter for tr in all_traces]))
            return col_iters * flop_col, col_iters / 64.0 * instr_col
        flop_d, instr_d = work(mask)
        flop_a, instr_a = work(np.ones(len(lengths), dtype=bool))
        alg_all = float(np.mean([algorithmic_bytes(tr, lengths, p, args.nmf_iter) for tr in all_traces]))
        # ALL kernels' work over the sweep: the longer of the dominant launch and the first-launch-to-last-end span (they agree
        # when the dominant kernel is the last to end, which is how the classes are scheduled; if they ever do not, the span is right)
        kernel_avg_ms = avg_ms
        avg_ms = max(avg_ms, span_ms)
        tflops = flop_a / (avg_ms * 1e-3) / 1e12
        simd_cycles = lambda ms: ms * 1e-3 * 2.4e9 * 256 * 4
        issue_peak = FP64_VECTOR_PEAK_TFLOPS * 4.0 / FP64_ISSUE_CYCLES_1WAVE
        traffic, tinfo = pmc_traffic(config, name_d, int(mask.sum())) if (world == 1 and n_genes == cfg['n_genes']) else (None, {'refused': 'not the profiled shard'})
        traffic_o = {eng.dev.class_kernel_name(c): (pmc_traffic(config, eng.dev.class_kernel_name(c), int(cls_mask[c].sum()))[0]
                                                    if traffic is not None else None) for c in others}
        traffic_all = (traffic + sum(traffic_o.values())) if (traffic is not None and all(v is not None for v in traffic_o.values())) else None
        out['roofline'] = {
            'bound': 'fp64_valu', 'achieved': tflops, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tflops / FP64_VECTOR_PEAK_TFLOPS,
            'what': 'fp64 vector work of ALL genes (every class kernel) / average launch duration of the dominant kernel: the '
                    'class kernels are launched back to back on their own streams, the dominant one is the last to end, so its '
                    'launch spans the whole sweep and the other kernels run INSIDE that window',
            'kernel': name_d, 'avg_launch_ms': kernel_avg_ms, 'sweep_ms': avg_ms, 'launches_timed': len(kernel_ms),
            'genes_in_kernel': int(mask.sum()), 'split_length': split, 'pair_length': tiny,
            'concurrent_kernels': [{'kernel': eng.dev.class_kernel_name(c), 'genes': int(cls_mask[c].sum()),
                                    'avg_launch_ms': cls_ms[c]} for c in others],
            'sweep_span_ms': span_ms,
            'dominant_kernel_own_work': {'fp64_tflops': flop_d / (kernel_avg_ms * 1e-3) / 1e12,
                                         'frac': flop_d / (kernel_avg_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                         'note': 'only the dominant kernel\'s genes over its launch duration (it has the chip to itself '
                                                 'for only part of that time)'},
            'issue_ceiling': {'peak': issue_peak, 'frac': tflops / issue_peak,
                              'what': 'one wave per SIMD issues every vector instruction of the column body at {0} cycles, not 4, whatever '
                                      'its kind (tools/ubench/instr_cost.hip; profiles/round3/ubench_instr_cost.txt)'
                                      .format(FP64_ISSUE_CYCLES_1WAVE)},
            'valu_issue_slots': {'wave_instructions': instr_a, 'slots_frac': instr_a * 4.0 / simd_cycles(avg_ms),
                                 'what': 'fp64 wave-instructions of the passes x 4 cycles / (sweep time x 2.4 GHz x 1024 SIMDs)'},
            'flop_per_column_iteration': flop_col, 'raw_count_units': raw_units,
            'power_steps_per_solve': power_steps_stats(all_traces, args.nmf_iter),
            'stream_read_ceiling_gbps': stream_gbps,
            'traffic': traffic_all, 'traffic_info': tinfo,
            'traffic_per_kernel': dict({name_d: traffic}, **traffic_o),
            'traffic_rate_gbps': (traffic_all / (avg_ms * 1e-3) / 1e9) if traffic_all else None,
            'traffic_frac_of_hbm_peak': (traffic_all / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_all else None,
            'hbm_algorithmic': {'bytes_per_sweep': alg_all, 'rate_gbps': alg_all / (avg_ms * 1e-3) / 1e9, 'hbm_peak_gbps': HBM_PEAK_GBPS,
                                'ratio_to_hbm_peak': alg_all / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                'what': 'SURVEY 8(d) algorithmic bytes (fp32 x and lambda re-streamed every inner iteration) / sweep time; '
                                        'NOT a roofline fraction: the kernels keep x + lambda on chip (registers + LDS), so these bytes are '
                                        'mostly never moved'},
            'note': 'rank-0 shard; HIP events on the library streams around each launch'}
    else:
        # config 4: after the row-maxima shortcut an outer iteration reads only the sampled columns; the kernel that
        # streams HBM is the initial ratio-SVD pass over the whole transcripts (nmf.py:109-121, :522-525); the kernel that
        # takes most of the time is the iteration kernel, a chain of tiny dependent eigen-solves per gene
        init_avg = float(np.mean(init_ms))
        avg_ms = float(np.mean(kernel_ms))
        alg_init = 8.0 * p * float(lengths.sum())                           # init pass 8 p L_g per gene
        sampled = float(np.mean([tr[:, 0].astype(np.float64).sum() for tr in all_traces]))
        calls = float(np.mean([tr[:, 1].astype(np.float64).sum() for tr in all_traces]))
        solves = calls * (args.nmf_iter + 1)                                # nmf.py:90-101: T + 1 rank-one approximations per nmf() call
        steps_pw = float(np.mean([tr[:, 7].astype(np.float64).sum() for tr in all_traces]))
        traffic4, tinfo4 = pmc_traffic(config, eng.dev.init_kernel_name(), n_genes) if (world == 1 and n_genes == cfg['n_genes']) else (None, {'refused': 'not the profiled shard'})
        out['roofline'] = {
            'bound': 'hbm', 'achieved': alg_init / (init_avg * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': alg_init / (init_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'traffic': traffic4, 'traffic_info': tinfo4,
            'kernel': eng.dev.init_kernel_name(), 'genes_in_kernel': n_genes, 'avg_launch_ms': init_avg, 'launches_timed': len(init_ms),
            'algorithmic_bytes_per_launch': alg_init,
            'stream_read_ceiling_gbps': stream_gbps,
            'frac_of_stream_read_ceiling': (alg_init / (init_avg * 1e-3) / 1e9 / stream_gbps) if stream_gbps else None,
            'row_maxima_kernel': {'kernel': 'k_row_max', 'ms': rowmax_ms, 'bytes': 4.0 * p * float(lengths.sum()),
                                  'gbps': (4.0 * p * float(lengths.sum()) / (rowmax_ms * 1e-3) / 1e9) if rowmax_ms > 0 else None,
                                  'frac_of_hbm_peak': (4.0 * p * float(lengths.sum()) / (rowmax_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if rowmax_ms > 0 else None,
                                  'what': 'once per upload (outside the step): one read of the packed coverage for the p row maxima of every gene'},
            'share_of_step': {'initial_pass_ms': init_avg, 'iteration_kernel_ms': avg_ms * args.iters, 'step_ms': dt / steps * 1e3},
            'iteration_kernel': iteration_kernel_bound(eng, name0, avg_ms, len(kernel_ms), sampled, calls, solves, steps_pw, lengths, p, rate),
            'shortcut': 'max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i): per-iteration full-length scans removed; without it SURVEY 8(d) '
                        'counts 8 p L_g per gene and outer iteration = {0:.3e} B per launch'.format(alg_init),
            'note': 'rank-0 shard; HIP events on the library stream'}
    out['setup'] = {'synth_s': t_gen, 'upload_s': t_up}
    if args.dump_traces:                                                # per-gene counters of the last step (tools/partition_study.py)
        np.savez_compressed(args.dump_traces + '.' + config, lengths=lengths, gene_ids=np.asarray(my_genes),
                            traces=np.stack([tr[:, :8] for tr in eng.traces]), class_ms=np.asarray(eng.class_ms))
    out['parity'] = parity_check(eng, pick, cfg, p, my_genes, lengths, split, args.nmf_iter, rate) if pick is not None else None
    final_scale = np.copy(eng.scale_factors)
    try:
        eng.dev.close()                                                 # free the shard before the follow-up runs open their own handles
    except Exception:
        pass
    if world > 1 and args.sharded_api:
        out['end_to_end_sharded'] = dict(end_to_end_sharded(ctx, cfg, p, n_genes, args, rate),
                                         pieces_of_the_timed_path={'upload_s': t_up, 'step_s': dt / steps})
    if world == 1 and n_cpu > 0:
        out['cpu_baseline'], ref, covs, rd = cpu_baseline(cfg, config, p, args.nmf_iter, args.iters, rate, n_cpu, n_single)
        if out['parity'] is not None:
            out['parity']['chain'] = chain_parity(ref, covs, rd, args.nmf_iter, args.iters, rate, local_rank)
        del ref, covs, rd
    else:
        out['cpu_baseline'] = None
    if keep_packed and out['parity'] is not None and n_genes == cfg['n_genes']:
        out['parity']['tie_sensitive'] = tie_sensitive_genes(packed, lengths, p, final_scale)
    if world == 1 and config == 'c2' and out['parity'] is not None and n_cpu > 0:
        out['parity']['pileup'] = pileup_parity(local_rank, p, final_scale, args.nmf_iter, 384)
    if keep_packed and want_e2e:
        out['end_to_end'] = end_to_end(packed, lengths, p, reads, args.iters, args.nmf_iter, rate, local_rank)
    return out


def iteration_kernel_bound(eng, name, avg_ms, launches, sampled, calls, solves, steps_pw, lengths, p, rate):
    """
    Config 4's dominant kernel (gen_rows::k_baseline_gen): per nmf() call T + 1 dependent eigen-solves of an n x n matrix
    (n <= 10), each a chain of MFMA squarings / reductions with nothing else of the gene to overlap -- a latency chain per
    wave.  Bound: solves per launch x warm latency of one inner iteration of a single wave / waves the chip keeps in flight.
    """
    d = {'kernel': name, 'avg_launch_ms': avg_ms, 'launches_timed': launches, 'active_columns_per_launch': sampled,
         'nmf_calls_per_launch': calls, 'eigen_solves_per_launch': solves, 'power_steps_per_launch': steps_pw,
         'bytes_needed_per_launch': 4.0 * p * float(np.ceil(lengths / float(rate)).sum()),
         'note': 'row maxima once per upload + only the sampled columns are read (SURVEY 8(d) shortcut, declared): bound by the '
                 'latency of ~100 tiny dependent eigen-solves per nmf() call, not by HBM'}
    path = os.path.join(ROOT, 'profiles', ROUND, 'rows_inner_iteration_cycles.json')
    try:
        with open(path) as f:
            u = json.load(f)
        cyc = float(u['cycles_per_inner_iteration_one_wave'])
        waves = float(u['waves_in_flight'])
        bound_ms = solves * cyc / 2.4e9 / waves * 1e3
        d['latency_bound'] = {'bound': 'dependent-chain latency', 'cycles_per_inner_iteration_one_wave': cyc, 'waves_in_flight': waves,
                              'bound_ms': bound_ms, 'frac': bound_ms / avg_ms, 'source': os.path.relpath(path, ROOT)}
    except (OSError, ValueError, KeyError):
        d['latency_bound'] = None
    return d


# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    args.gpus = world

    # stdout carries ONE JSON line: libraries that print there (RCCL's version banner at communicator creation) are sent to
    # stderr for the life of the rank; the line itself goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from degnorm_amd.nmf_mpi import TorchComm, LocalComm

    rehearsal = args.backend == 'gloo'
    if rehearsal:
        local_rank = 0                                                  # all ranks share the one GPU
    torch.cuda.set_device(local_rank)
    under_launcher = 'TORCHELASTIC_RUN_ID' in os.environ or 'WORLD_SIZE' in os.environ
    comm, rccl, rccl_ranks, pg = LocalComm(), None, None, None
    if under_launcher:
        if rehearsal:
            dist.init_process_group('gloo')
            comm = TorchComm(device='cpu')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            comm = TorchComm(device='cuda:{0}'.format(local_rank))
        pg = dist
    elif not args.no_rccl:
        # plain `python bench.py` at N = 1: the per-sample all-reduce still goes through RCCL (a one-rank process group), so the
        # N = 1 line pays what every rank of an N > 1 run pays
        try:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:{0}'.format(_free_port()), world_size=1, rank=0,
                                    device_id=torch.device('cuda', local_rank))
            comm = TorchComm(device='cuda:{0}'.format(local_rank))
            pg = dist
        except Exception as e:                                          # no RCCL on this box: the all-reduce of one rank is the identity
            rccl = {'backend': None, 'error': repr(e)[:200]}
    if pg is not None:
        rccl = {'backend': dist.get_backend(), 'rccl_ranks': dist.get_world_size(),
                'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()), 'rehearsal_on_one_gpu': rehearsal,
                'self_launched': bool(os.environ.get('DN_BENCH_PARENT'))}
        rccl_ranks = dist.get_world_size() if dist.get_backend() == 'nccl' else None
    ctx = {'rank': rank, 'world': world, 'local_rank': local_rank, 'comm': comm, 'dist': pg, 'rehearsal': rehearsal,
           'rccl': rccl, 'rccl_ranks': rccl_ranks}

    n_cpu = args.cpu_sample if args.cpu_sample >= 0 else (768 if args.config == 'c2' else 2048)
    n_single = args.cpu_single if args.cpu_single >= 0 else (64 if args.config == 'c2' else 512)
    out = measure(args.config, args, ctx, args.steps, args.warmup, args.parity_genes, n_cpu, n_single, not args.no_end_to_end)
    if rank == 0 and world == 1 and args.config == 'c2' and not args.no_also and args.genes <= 0:
        # BASELINE configs[3] in the same process, after the config-2 clock has stopped (its own generation, upload, warm-up, clock)
        sub = measure('c4', args, ctx, 3, 2, args.parity_genes, 2048 if n_cpu > 0 else 0, 512 if n_single > 0 else 0, False)
        out['also'] = {'config 4': {k: sub[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'config', 'roofline',
                                                        'parity', 'cpu_baseline', 'setup')}}
    if pg is not None:
        if world > 1:
            comm.Barrier()                                              # rank 0's post-clock checks are over: everybody leaves together
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    os.close(result_fd)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2', choices=['c2', 'c4'])
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=2)    # the first step after the first-ever one still carries ~18 ms of one-time host cost
    ap.add_argument('--genes', type=int, default=0, help='total genes (default: the configuration\'s: 20000 / 50000)')
    ap.add_argument('--iters', type=int, default=5, help='outer DegNorm iterations per step')
    ap.add_argument('--nmf-iter', type=int, default=100)
    ap.add_argument('--cpu-sample', type=int, default=-1, help='genes in the CPU-baseline sample (0 = skip; default 768 / 2048)')
    ap.add_argument('--cpu-single', type=int, default=-1, help='genes of that sample also run on ONE thread (0 = skip; default 64 / 512)')
    ap.add_argument('--parity-genes', type=int, default=160, help='genes in the post-clock parity sample (0 = skip)')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the GeneNMFOA.fit() end-to-end timing (config 2, N = 1)')
    ap.add_argument('--no-also', action='store_true', help='skip the config-4 measurement appended to the default config-2 line')
    ap.add_argument('--no-rccl', action='store_true', help='N = 1 without torchrun: do not open a one-rank RCCL process group')
    ap.add_argument('--redeal', action='store_true',
                    help='N > 1: before the warm-up, one untimed run re-deals the genes from the measured cost of its first outer iteration')
    ap.add_argument('--sharded-api', action='store_true',
                    help='N > 1: after the clock, also run the reference-signature API run_gene_nmfoa_mpi on the whole configuration '
                         '(rank 0 holds the float64 dict) and report scatter / run / gather per rank (`end_to_end_sharded`)')
    ap.add_argument('--torch-collective', action='store_true',
                    help='run the per-iteration all-reduce through torch.distributed on the library\'s device buffer (round 3) instead of '
                         'inside the library (dn_comm_*, the default with the nccl backend)')
    ap.add_argument('--dump-traces', default='', help='write the per-gene device counters of the last step to FILE.<config>.npz')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='collective backend of the ranks: nccl = RCCL (the measured path); gloo only to rehearse N > 1 on ONE GPU '
                         '(every rank on device 0; the line is then marked "rehearsal")')
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, argv)                            # parent: nothing below this line touches torch or the GPU
    run_rank(args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
