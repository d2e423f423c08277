"""
Test-only stand-in for degnorm_amd._lib.Device backed by the CPU oracle, so that the host logic above the
C ABI (sharding, all-reduce algebra, gather, result writing) can be exercised on machines without a GPU.
It is the CHECKER wearing the device's interface; nothing in the product imports it.

    from _oracle_device import use_oracle_device
    use_oracle_device()          # degnorm_amd._lib.Device now builds OracleDevice objects (this process only)
"""
import numpy as np

from oracle import oracle as orc


class OracleDevice(object):
    def __init__(self, device=0):
        self.n = 0
        self.p = 0
        self.lengths = None
        self.inexact = 0
        self._covs = None
        self._est = None

    def hint_downsample(self, rate):
        return self

    def upload(self, cov_mats, n_threads=0):
        self._covs = [np.ascontiguousarray(c, dtype=np.float64) for c in cov_mats]
        self.n, self.p = len(self._covs), self._covs[0].shape[0]
        self.lengths = np.array([c.shape[1] for c in self._covs], dtype=np.int64)
        return self

    def upload_packed(self, packed, lengths, p):
        covs, o = [], 0
        for L in lengths:
            covs.append(np.asarray(packed[o:o + p * int(L)], dtype=np.float64).reshape(p, int(L)))
            o += p * int(L)
        return self.upload(covs)

    def ratio_svd_sums(self):
        return orc.ratio_svd_batch(self._covs)

    def baseline_iteration(self, scale, nmf_iter=100, bins=20, min_high_coverage=50, downsample_rate=1,
                           skip_baseline_selection=False, want_estimates=False, ds_start=None, want_trace=True):
        prm = orc._Params(int(nmf_iter), int(bins), int(min_high_coverage), int(downsample_rate),
                          int(bool(skip_baseline_selection)))
        rho, flags, trace, est = orc.baseline_batch(self._covs, scale, prm, ds_start=ds_start,
                                                    want_estimates=want_estimates)
        self._est = est
        return rho, flags, trace

    def fetch_estimates(self):
        return self._est

    def last_kernel_ms(self):
        return 0.0

    def close(self):
        pass


def use_oracle_device():
    """Point the product's device constructor at the stand-in (test processes only; the product has no such switch)."""
    import degnorm_amd._lib as L
    L.Device = OracleDevice
