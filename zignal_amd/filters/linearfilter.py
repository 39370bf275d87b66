"""The slice of zignal's `Filter` that the MLS emphasis / de-emphasis step uses
(zignal/filters/linearfilter.py:28-143): coefficient holder, `filter_samples`
(= scipy.signal.lfilter along axis 0) and the minimum-phase test MLS asserts on
(zignal/measure/mls.py:365-369). `filter_samples` runs on the device (csrc/iir.hip:
lfilter's transposed-direct-form-II recurrence, block-parallel) for
orders up to 8, which covers every filter zignal ships; the MLS classes call the same
kernels directly (zignal_amd/measure/mls.py). The pole/zero tests are host work.
"""
import numpy as np
import scipy.signal


class Filter(object):
    def __init__(self, B=None, A=None, fs=96000):
        assert fs > 0
        self.fs = fs
        self.set_coefficients(B=B, A=A)

    def set_coefficients(self, B=None, A=None):
        self._B = np.array((1,)) if B is None else np.array(B)
        self._A = np.array((1,)) if A is None else np.array(A)
        assert len(self._B) != 0
        assert len(self._A) != 0

    def get_coefficients(self):
        return self._B, self._A

    def normalise(self):
        a0 = self._A[0]
        self._B = self._B / a0
        self._A = self._A / a0

    def filter_samples(self, samples):
        """scipy.signal.lfilter(B, A, samples, axis=0) (zignal/filters/linearfilter.py:62-63), computed on the GPU.
        Like every compute call of this package it fails loudly without the HIP library or a GPU; orders above 8
        (none in zignal) are refused by the library."""
        from ..device import IirFilter
        f = IirFilter(self._B, self._A)
        try:
            return f.filter_host(np.asarray(samples))
        finally:
            f.close()

    def _zeros_poles(self):
        z, p, _ = scipy.signal.tf2zpk(self._B, self._A)
        return z, p

    def is_stable(self):
        return bool(np.all(np.abs(self._zeros_poles()[1]) < 1.0))

    def is_minimum_phase(self):
        z, p = self._zeros_poles()
        return bool(np.all(np.abs(p) < 1.0) and np.all(np.abs(z) < 1.0))

    def __repr__(self):
        return "Filter(B=%s, A=%s, fs=%s)" % (list(self._B), list(self._A), self.fs)
