"""Timing of the device coverage assembly on a chromosome-scale synthetic input."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from scipy import sparse
from degnorm_amd.coverage_merge import assemble_chrom_packed
chrom_len = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_genes = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(prefix='dn_asm_')
sids = ['s%d' % i for i in range(p)]
starts = np.sort(rng.integers(1000, chrom_len - 20000, size=n_genes))
rows = []
for g, st in enumerate(starts):
    pos = int(st)
    ex = []
    for _ in range(int(rng.integers(2, 9))):
        ln = int(rng.integers(80, 700)); ex.append((pos, pos + ln - 1)); pos += ln + int(rng.integers(100, 1500))
    for a, b in ex:
        rows.append(dict(chr='chrA', gene='g%05d' % g, gene_start=ex[0][0], gene_end=ex[-1][1], start=a, end=b))
exon_df = pd.DataFrame(rows)
for s in sids:
    nnz = chrom_len // 50
    idx = np.unique(rng.integers(0, chrom_len, size=nnz)).astype(np.int32)
    val = rng.poisson(20, size=idx.size) + 1
    os.makedirs(os.path.join(d, s))
    sparse.save_npz(os.path.join(d, s, 'chrom_coverage_%s_chrA.npz' % s),
                    sparse.csr_matrix((val, idx, np.array([0, idx.size])), shape=(1, chrom_len)), compressed=False)
t0 = time.time()
genes, packed, lengths, ms = assemble_chrom_packed(d, sids, exon_df, verbose=False)
wall = time.time() - t0
gathered = packed.nbytes
touched = p * (chrom_len * 4 + 2 * gathered / p)          # memset of the dense vector + gather read/write per sample
print('chrom_len %.0e, %d samples, %d genes, %.1f MB packed: device %.2f ms (%.0f GB/s over memset + gather traffic), wall %.2f s (npz load + interval build + H2D/D2H)'
      % (chrom_len, p, len(genes), gathered / 1e6, ms, touched / ms / 1e6, wall))
