"""bench.py's post-clock accounting helpers that need no GPU: the tie count of get_high_coverage_idx and the parity sample."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_tie_sensitive_genes_counts_columns_on_the_threshold():
    import bench
    p = 3
    a = np.array([[100., 10., 50., 9.], [5., 3., 2., 1.], [1., 1., 1., 1.]], dtype=np.float32)      # column 1: 10 x 10 == 100 -> on the threshold
    b = np.array([[100., 11., 50.], [5., 3., 2.], [1., 1., 1.]], dtype=np.float32)                  # no column at 10
    c = np.array([[40., 4., 4., 39.], [7., 1., 1., 1.], [2., 2., 2., 2.]], dtype=np.float32)        # two columns at max / 10
    packed = np.concatenate([m.ravel() for m in (a, b, c)])
    out = bench.tie_sensitive_genes(packed, [4, 3, 4], p, np.ones(p))
    assert out['genes'] == 2 and out['columns'] == 3 and out['of_genes'] == 3
    # a scale factor that is not a power of two keeps exact ties of the SAME row on the threshold (the quotient is monotone and the
    # two sides round alike), which is why such genes flip on the last bit of s
    out = bench.tie_sensitive_genes(packed, [4, 3, 4], p, np.array([1.1, 1.0, 1.0]))
    assert out['genes'] >= 1


def test_parity_sample_spans_all_lengths():
    import bench
    lengths = np.arange(200, 5000, 7)
    pick = bench.parity_sample(lengths, 160)
    assert len(pick) <= 160 and len(np.unique(pick)) == len(pick)
    assert lengths[pick].min() == lengths.min() and lengths[pick].max() == lengths.max()
    assert len(bench.parity_sample(lengths[:10], 160)) == 10


def test_algorithmic_bytes_formula():
    """bytes_g = 4 p [ 2 L_g + sum_k n_{g,k} (3 T + 3) ]."""
    import bench
    trace = np.zeros((2, 8), dtype=np.int32)
    trace[:, 2] = [1000, 0]
    lengths = np.array([500., 300.])
    assert bench.algorithmic_bytes(trace, lengths, 10, 100) == 4 * 10 * (2 * 500 + 1000 * 303) + 4 * 10 * (2 * 300)
    assert bench.algorithmic_bytes(trace, lengths, 10, 100, mask=np.array([False, True])) == 4 * 10 * 600
