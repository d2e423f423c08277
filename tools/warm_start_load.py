"""Load time of a warm-start directory with and without the packed float32 side-car (degnorm_amd/warm_start.py).
BASELINE config 5's shape: 6 samples, ~2 000 genes over three chromosomes, L ~ U[200, 5000].  CPU only.
    python tools/warm_start_load.py [n_genes]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import _fixtures
from degnorm_amd.warm_start import load_from_previous, write_sidecars

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
d = tempfile.mkdtemp(prefix='dn_ws_')
try:
    _fixtures.write_warm_start_dir(d, seed=5, n_genes=n, p=6, l_min=200, l_max=5000)
    pk = sum(os.path.getsize(os.path.join(d, c, f)) for c in os.listdir(d) if os.path.isdir(os.path.join(d, c)) for f in os.listdir(os.path.join(d, c)))
    def timed(use):
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            dat = load_from_previous(d, use_sidecar=use)
            tot = sum(float(m[0, 0]) for m in dat['gene_cov_dict'].values())        # touch every matrix
            best = min(best, time.time() - t0)
        return best, len(dat['gene_cov_dict'])
    t_pkl, ng = timed(False)
    t0 = time.time(); write_sidecars(d); t_write = time.time() - t0
    sc = sum(os.path.getsize(os.path.join(d, c, f)) for c in os.listdir(d) if os.path.isdir(os.path.join(d, c)) for f in os.listdir(os.path.join(d, c)) if '.f32.' in f)
    t_sc, _ = timed(True)
    print('{0} genes x 6 samples: pickles {1:.1f} MB, side-cars {2:.1f} MB (written once in {3:.2f} s)'.format(ng, pk / 1e6, sc / 1e6, t_write))
    print('load_from_previous, best of 3 (page cache warm): pickles {0:.3f} s, float32 side-car {1:.3f} s ({2:.1f}x)'.format(t_pkl, t_sc, t_pkl / t_sc))
finally:
    shutil.rmtree(d)
