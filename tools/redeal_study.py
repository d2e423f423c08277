"""
What re-dealing after the first outer iteration buys on N ranks, from the per-gene counters of a 1-GPU run (bench.py --dump-traces
FILE writes FILE.c2.npz: lengths, gene ids, traces of the last step's outer iterations): the cost-balanced deal of the lengths
(utils.partition_by_cost, what every rank starts with), the MEASURED cost of outer iteration 1 (utils.measured_gene_cost), the few
moves of utils.rebalance_moves, and how level the LATER iterations then are (the point of it: iteration 1 predicts them).
usage: python tools/redeal_study.py FILE.c2.npz [ranks ...]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from degnorm_amd.utils import partition_by_cost, measured_gene_cost, rebalance_moves, predicted_gene_cost

d = np.load(sys.argv[1])
L, tr = d['lengths'], d['traces']
ranks = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
p = 10
cls = (int(d['split_len']), int(d['tiny_len'])) if 'split_len' in d.files else None
print('%d genes, %d outer iterations of counters; class boundaries %s' % (len(L), len(tr), cls))
for n in ranks:
    parts = partition_by_cost(L, n, p=p, class_lengths=cls)
    owner = np.zeros(len(L), dtype=np.int64)
    for r, q in enumerate(parts):
        owner[q] = r
    cost1 = measured_gene_cost(tr[0], L, p, cls)
    moves = rebalance_moves(owner, cost1, n)
    owner2 = owner.copy()
    for g, s, t in moves:
        owner2[g] = t
    line = []
    for i in range(len(tr)):
        c = measured_gene_cost(tr[i], L, p, cls)
        a = np.bincount(owner, weights=c, minlength=n); b = np.bincount(owner2, weights=c, minlength=n)
        line.append('it %d: %.4f -> %.4f' % (i + 1, a.max() / a.mean(), b.max() / b.mean()))
    pc = predicted_gene_cost(L, p, 1, cls)
    a = np.bincount(owner, weights=pc, minlength=n)
    print('%d ranks: predicted-from-length max/mean %.4f; measured max/mean before -> after moving %d genes (%.1f MB of fp32 coverage): %s'
          % (n, a.max() / a.mean(), len(moves), 4e-6 * p * sum(int(L[g]) for g, _, _ in moves), '; '.join(line)))
