"""
Config 4's iteration kernel (gen_rows::k_baseline_gen, one wavefront per gene): how long ONE wave alone on a SIMD needs for an
inner NMF-OA iteration (update + n x n Gram products + register reduce-scatter + MFMA eigen-solve: one dependent chain), from a
launch with 1 024 genes = 1 024 workgroups = one wave per SIMD.  With W waves per SIMD in flight the kernel cannot finish the
launch's inner iterations faster than  solves x t1 / (waves in flight)  -- the bound bench.py's c4 line quotes
(`roofline.iteration_kernel.latency_bound`).  Writes profiles/<round>/rows_inner_iteration_cycles.json.
usage (GPU box): python tools/rows_latency.py [out.json]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from degnorm_amd import synth, _lib

cfg = synth.CONFIGS['c4']
T, rate, n = 100, 500, 1024
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
dev = _lib.Device(0)
dev.hint_downsample(rate)
dev.upload_packed(packed, lengths, cfg['p'])
ds = np.random.RandomState(1).randint(0, rate, size=n).astype(np.int64)
best = None
for rep in range(4):
    rho, flags, tr = dev.baseline_iteration(np.ones(cfg['p']), nmf_iter=T, min_high_coverage=2, downsample_rate=rate, ds_start=ds)
    ms = dev.last_kernel_ms()
    best = ms if best is None else min(best, ms)
calls = tr[:, 1].astype(float)
# every workgroup holds one gene: the launch lasts as long as the gene with the most nmf() calls (each T + 1 inner iterations)
t1 = best * 1e-3 * 2.4e9 / (calls.max() * (T + 1))
per_cu = dev.lib.dn_class_kernel_ms  # (unused; keeps the symbol referenced)
out = {'kernel': dev.class_kernel_name(0), 'genes': n, 'launch_ms': best, 'max_nmf_calls_of_a_gene': int(calls.max()),
       'mean_active_columns_per_call': float(tr[:, 2].sum() / max(1.0, calls.sum())),
       'cycles_per_inner_iteration_one_wave': t1, 'clock_ghz': 2.4, 'waves_in_flight': 256 * 4 * 3,
       'how': '1 024 genes on 1 024 single-wave workgroups (one wave per SIMD): launch time / (most nmf() calls of a gene x (T + 1)); '
              'the product kernel keeps 3 waves per SIMD (launch bounds), 3 072 on the chip'}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
