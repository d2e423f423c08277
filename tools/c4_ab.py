"""A/B of library variants in BASELINE config 4's regime (p = 50, take-every 500) on ONE generated slice.
usage (GPU box): python tools/c4_ab.py <n_genes> <variant> [<variant> ...]      variants: build_variants/lib_<variant>.so ('tree' = the product library)
The parent generates the slice once (memory-mapped file in /dev/shm); every variant runs in its own child process (one library per
process) and reports the device times of the initial pass and of the iteration kernel plus the DI matrix, compared with the first variant's."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

SHM = '/dev/shm/dn_c4_slice'


def child(n):
    from degnorm_amd.nmf_mpi import ShardedNMFOA
    packed = np.load(SHM + '.packed.npy', mmap_mode='r')
    meta = np.load(SHM + '.meta.npz')
    eng = ShardedNMFOA(degnorm_iter=5, nmf_iter=100, downsample_rate=500)
    eng.load_packed(packed, meta['lengths'], 50, meta['reads'])
    best = None
    for rep in range(3):
        t0 = time.time()
        eng.initialize()
        init_ms = eng.dev.last_init_ms()
        for i in range(5):
            eng.iterate(i)
        eng.fetch_state()
        dt = time.time() - t0
        row = (dt, init_ms, float(np.mean(eng.kernel_ms)))
        best = row if best is None or row[0] < best[0] else best
    np.save(SHM + '.rho.' + os.environ.get('DN_VARIANT', 'x') + '.npy', eng.rho)
    print('%-10s run %.1f ms  init kernel %.2f ms  iteration kernel %.2f ms  -> %.0f genes/s' % (
        os.environ.get('DN_VARIANT'), best[0] * 1e3, best[1], best[2], n / best[0]), flush=True)


if __name__ == '__main__':
    n = int(sys.argv[1])
    if os.environ.get('DN_VARIANT'):
        child(n)
        sys.exit(0)
    from degnorm_amd import synth
    cfg = synth.CONFIGS['c4']
    t0 = time.time()
    packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
    np.save(SHM + '.packed.npy', packed)
    np.savez(SHM + '.meta.npz', lengths=lengths, reads=reads)
    del packed
    print('slice: %d genes, generated in %.1f s' % (n, time.time() - t0), flush=True)
    ref = None
    try:
        for v in sys.argv[2:]:
            env = dict(os.environ, DN_VARIANT=v)
            if v != 'tree':
                env['DN_LIB_PATH'] = os.path.join(ROOT, 'build_variants', 'lib_%s.so' % v)
            subprocess.run([sys.executable, os.path.abspath(__file__), str(n)], env=env, check=True)
            rho = np.load(SHM + '.rho.%s.npy' % v)
            if ref is None:
                ref = rho
            else:
                print('           max |d rho| vs %s: %.2e' % (sys.argv[2], float(np.abs(rho - ref).max())), flush=True)
    finally:
        for f in os.listdir('/dev/shm'):
            if f.startswith('dn_c4_slice'):
                os.remove(os.path.join('/dev/shm', f))
