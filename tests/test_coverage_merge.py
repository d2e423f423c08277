"""
SURVEY 8(f-3): coverage-matrix assembly.  Golden tests/golden/merge.npz = the reference's merge_chrom_coverage
(reads_coverage_merge.py:167-372) on the synthetic directory of tests/_fixtures.py write_chrom_coverage_dir.
The interval logic runs on CPU; the device gather needs the GPU.  Coverage is integer counts: comparisons are exact.
"""
import numpy as np
import pytest

from conftest import golden
from degnorm_amd import synth  # noqa: F401
import _fixtures
from degnorm_amd.coverage_merge import gene_intervals, merge_chrom_coverage, assemble_chrom_packed


def test_gene_order_and_exon_unions_match_reference(tmp_path):
    G = golden('merge')
    sample_ids, exon_df = _fixtures.write_chrom_coverage_dir(str(tmp_path / 'cov'))
    assert sample_ids == list(G['sample_ids'])
    genes, ivs = gene_intervals(exon_df)
    assert genes == list(G['genes'])                                   # reference order: sorted by gene_end (:264-267)
    np.testing.assert_array_equal([sum(b - a for a, b in iv) for iv in ivs], G['lengths'])
    for iv in ivs:                                                     # merged, ascending, disjoint
        assert all(iv[k][1] < iv[k + 1][0] for k in range(len(iv) - 1))


@pytest.mark.gpu
def test_device_assembly_matches_reference_exactly(tmp_path):
    G = golden('merge')
    d = str(tmp_path / 'cov')
    sample_ids, exon_df = _fixtures.write_chrom_coverage_dir(d)
    out = merge_chrom_coverage(d, sample_ids, exon_df, verbose=False)
    assert list(out.keys()) == list(G['genes'])
    o = 0
    for g, L in zip(G['genes'], G['lengths']):
        m = out[g]
        assert m.shape == (len(sample_ids), int(L)) and m.dtype == np.float64
        np.testing.assert_array_equal(m.reshape(-1), G['flat'][o:o + m.size])
        o += m.size
    assert not out[G['genes'][0]][2].any()                             # sample 2 has no file: imputed zeros (:309-316)
    # straight into the NMF-OA core without a host dictionary
    from degnorm_amd import _lib
    genes, packed, lengths, ms = assemble_chrom_packed(d, sample_ids, exon_df, verbose=False)
    dev = _lib.Device(0)
    dev.upload_packed(packed, lengths, len(sample_ids))
    est, cov, status = dev.ratio_svd_sums()
    np.testing.assert_allclose(cov, np.vstack([out[g].sum(axis=1) for g in genes]), rtol=1e-14)
    dev.close()
    # no coverage at all for a chromosome -> empty dict (:247-252); two chromosomes -> ValueError (:233-235)
    other = exon_df.copy()
    other['chr'] = 'chrNone'
    assert merge_chrom_coverage(d, sample_ids, other, verbose=False) == {}
    both = exon_df.copy()
    both.loc[0, 'chr'] = 'chrZ'
    with pytest.raises(ValueError, match='more than one chromosome'):
        merge_chrom_coverage(d, sample_ids, both, verbose=False)
