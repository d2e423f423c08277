"""A/B of library variants on the initial pass alone (gen::k_ratio_svd_mg) on one generated config-4 slice.
usage (GPU box): python tools/init_ab.py <n_genes> <variant> [...]      variants: build_variants/lib_<variant>.so ('tree' = the product library)
Every variant runs in its own child process; device time of the kernel (best of 4) and, against the first variant, the largest
relative difference of the clamped row sums (diagnostic knock-out builds give garbage there: only their time is of interest)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

SHM = '/dev/shm/dn_init_slice'


def child():
    from degnorm_amd import _lib
    packed = np.load(SHM + '.packed.npy', mmap_mode='r')
    meta = np.load(SHM + '.meta.npz')
    dev = _lib.Device(0)
    dev.hint_downsample(500)
    dev.upload_packed(packed, meta['lengths'], 50)
    best = 1e9
    for rep in range(4):
        est, cov, st = dev.ratio_svd_sums()
        best = min(best, dev.last_init_ms())
    np.save(SHM + '.est.' + os.environ['DN_VARIANT'] + '.npy', est)
    print('%-10s init kernel %.3f ms   (%.2f TB/s of algorithmic bytes)' % (os.environ['DN_VARIANT'], best, 2 * packed.nbytes / best * 1e-9), flush=True)


if __name__ == '__main__':
    if os.environ.get('DN_VARIANT'):
        child()
        sys.exit(0)
    n = int(sys.argv[1])
    from degnorm_amd import synth
    cfg = synth.CONFIGS['c4']
    packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
    np.save(SHM + '.packed.npy', packed)
    np.savez(SHM + '.meta.npz', lengths=lengths)
    print('slice: %d genes, %.2f GB' % (n, packed.nbytes / 1e9), flush=True)
    del packed
    ref = None
    try:
        for v in sys.argv[2:]:
            env = dict(os.environ, DN_VARIANT=v)
            if v != 'tree':
                env['DN_LIB_PATH'] = os.path.join(ROOT, 'build_variants', 'lib_%s.so' % v)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
            est = np.load(SHM + '.est.%s.npy' % v)
            if ref is None:
                ref = est
            else:
                with np.errstate(all='ignore'):
                    print('           max rel d(est_sums) vs %s: %.2e' % (sys.argv[2], float(np.nanmax(np.abs(est - ref) / (np.abs(ref) + 1e-300)))), flush=True)
    finally:
        for f in os.listdir('/dev/shm'):
            if f.startswith('dn_init_slice'):
                os.remove(os.path.join('/dev/shm', f))
