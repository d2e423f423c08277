"""
A/B of the hot loop's eigen-solvers on the Gram-matrix sequences of real nmf() calls (config-2 genes): cycles per warm solve at one
wave per SIMD, steps, and the distance of every returned eigenvector from numpy's eigh.  GPU box:
    python tools/ubench/solver_ab.py [n_genes]          (builds tools/ubench/libsolver_ab.so if missing)
"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from degnorm_amd import synth

P, T = 10, 100


def build():
    so = os.path.join(ROOT, 'tools', 'ubench', 'libsolver_ab.so')
    src = os.path.join(ROOT, 'tools', 'ubench', 'solver_ab.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DDN_P=%d' % P, '-DDN_NT=256',
                        '-I' + os.path.join(ROOT, 'degnorm_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'), '-o', so, src], check=True)
    return so


def gram_sequence(x):
    """the T + 1 Gram matrices of one nmf() call (nmf.py:78-107) with exact eigenvectors (numpy eigh)"""
    p, n = x.shape
    lam = np.zeros_like(x)
    c = 1.0 / np.sqrt(T)
    out, us = [], []
    a = x
    for t in range(T + 1):
        G = a @ a.T
        w, V = np.linalg.eigh(G)
        u = V[:, -1] * np.sign(V[:, -1].sum())
        out.append(G[np.tril_indices(p)]); us.append(np.concatenate([u, [w[-1]]]))
        s = u @ a
        lam = np.maximum(lam - c * (np.outer(u, s) - x), 0.0)
        a = x + lam
    return np.array(out), np.array(us)


def main():
    if '--build-only' in sys.argv:
        build(); return
    n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    lib = ctypes.CDLL(build())
    cfg = synth.CONFIGS['c2']
    tot = np.zeros(9); k = 0
    worst = np.zeros(2)
    for g in range(n_genes):
        cov, cls = synth.synth_gene(cfg['seed'], g, P, 200, 5000)
        x = cov[:, cov.max(0) > 0.1 * cov.max()].astype(np.float64)
        if x.shape[1] < 50:
            continue
        Gs, us = gram_sequence(x)
        Gs = np.ascontiguousarray(Gs)
        uo = np.zeros((2, T + 1, P + 1)); res = np.zeros(9)
        rc = lib.solver_ab(Gs.ctypes.data_as(ctypes.c_void_p), T + 1, uo.ctypes.data_as(ctypes.c_void_p), res.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        err = [np.abs(np.abs(uo[v, :, :P]) - np.abs(us[:, :P])).max() for v in range(2)]
        th = [abs(uo[v, -1, P] / us[-1, P] - 1.0) for v in range(2)]
        print('gene %3d class %d n %4d | mfma %6.0f cycles %5.2f steps err %.1e theta %.1e | dpp %6.0f cycles %5.2f steps err %.1e theta %.1e'
              % (g, cls, x.shape[1], res[0], res[1], err[0], th[0], res[2], res[3], err[1], th[1]), flush=True)
        tot += res; k += 1
        worst = np.maximum(worst, err)
    print('mean over %d genes: mfma %.0f cycles per warm solve, dpp %.0f cycles (%.2f steps); worst |u - eigh| mfma %.1e dpp %.1e'
          % (k, tot[0] / k, tot[2] / k, tot[3] / k, worst[0], worst[1]))
    print('dpp phases (cycles per solve): load %.0f, blind steps %.0f, first normalisation %.0f, looks %.0f, epilogue %.0f' % tuple(tot[4:] / k))


if __name__ == '__main__':
    main()
