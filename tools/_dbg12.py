import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from degnorm_amd import synth, _lib
from oracle import oracle
oracle.build()
p=int(sys.argv[1]) if len(sys.argv)>1 else 12
rng = np.random.default_rng(100 + p)
covs = [synth.synth_gene(9, g, p, 60, 900)[0] for g in range(10)]
covs += [rng.poisson(30, size=(p, L)).astype(float) for L in (2, 3, 5, 51, 64, 65, 257)]
covs.append(np.zeros((p, 40)))
covs.append(np.tile(np.arange(1, 301, dtype=float), (p, 1)))
scale = np.linspace(0.9, 1.2, p)
dev=_lib.Device(0)
for bins, T, mhc, skip in ((20, 12, 50, False), (5, 3, 2, False), (33, 1, 10, True)):
    dev.upload(covs)
    rho, flags, trace = dev.baseline_iteration(scale, nmf_iter=T, bins=bins, min_high_coverage=mhc, skip_baseline_selection=skip, want_estimates=True)
    prm = oracle.make_params(nmf_iter=T, bins=bins, min_high_coverage=mhc, skip_baseline_selection=skip)
    rho_o, flags_o, trace_o, est_o = oracle.baseline_batch(covs, scale, prm, want_estimates=True)
    for g in range(len(covs)):
        if not np.array_equal(trace[g,:7], trace_o[g,:7]) or not np.allclose(rho[g], rho_o[g], rtol=1e-8, atol=1e-10):
            print('cfg',(bins,T,mhc,skip),'gene',g,'L',covs[g].shape[1],'dev',trace[g,:8].tolist(),'orc',trace_o[g,:8].tolist(),'drho',float(np.abs(rho[g]-rho_o[g]).max()))
print('done', dev.class_kernel_name(0), dev.class_kernel_name(1), dev.class_kernel_name(2))
