"""
degnorm_amd -- MI355X-native NMF over-approximation core of DegNorm (hot path only).

    from degnorm_amd import GeneNMFOA            # mirror of degnorm.nmf.GeneNMFOA (reference nmf.py:10)
    from degnorm_amd import run_gene_nmfoa_mpi   # mirror of degnorm.nmf_mpi.run_gene_nmfoa_mpi (nmf_mpi.py:555)
"""
__version__ = '0.1.0'


def __getattr__(name):
    # lazy: importing the package (e.g. for degnorm_amd.synth) must not need the HIP library.
    if name in ('GeneNMFOA',):
        from . import nmf
        return getattr(nmf, name)
    if name in ('run_gene_nmfoa_mpi', 'save_results'):
        from . import nmf_mpi
        return getattr(nmf_mpi, name)
    raise AttributeError(name)
