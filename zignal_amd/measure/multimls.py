"""Several MLS stimuli at once: one output channel per tap set, every capture channel paired with every
stimulus at extraction — SURVEY.md §8(f) row 3.

The reference describes this and leaves it to the caller (zignal/measure/mls.py:338-343: "creating multiple
instances ... constructing a new Audio instance with all channels appended. To extract the impulse response we
then need to keep track of which channel is paired with which MLS since they should preferably be using
different taps to minimise cross talk"). Here it is a front end over one device plan per tap set: the capture
is uploaded once and every plan correlates all of its channels on the GPU, so the (output k -> input c) response
is `impulses[k][:, c]`. Each pair equals what `_MLS_base(N, taps_k).get_impulse(x[:, c])` returns.
"""
import numpy as np

from .. import _lib as _L
from ..audio import Audio
from ..device import DeviceBuffer, Plan, sync
from .mlstaps import TAPS


class MultiMLS(object):
    def __init__(self, N=None, taps=None, fs=96000, repeats=2, dtype='f64', device=None):
        """taps: a sequence of tap sets, one per output channel (all different), or an int K to take the first K
        sets of TAPS[N]."""
        assert N is not None, "Please specify MLS order"
        assert taps is not None, "Please specify feedback taps"
        assert repeats > 1, "at least two sequences are needed, (repeats=2)"
        if isinstance(taps, int):
            assert 1 <= taps <= len(TAPS[N]), "TAPS[%d] holds %d tap sets" % (N, len(TAPS[N]))
            taps = TAPS[N][:taps]
        taps = [tuple(t) for t in taps]
        assert len(taps) != 0, "taps are empty!"
        assert len(set(taps)) == len(taps), "every output needs its own tap set (cross talk otherwise)"
        self.N, self.L, self.fs, self.repeats = N, (2 ** N) - 1, fs, repeats
        self.taps = taps
        self._plans = [Plan(N, t, dtype=dtype, device=device) for t in taps]
        self.device = self._plans[0].device

    def __len__(self):
        return len(self._plans)

    def get_stimulus(self):
        """Audio with one channel per tap set: (repeats*L, K) float64, channel k = get_full_sequence of tap set k."""
        cols = [p.sequence(self.repeats)[:, 0] for p in self._plans]
        return Audio(fs=self.fs, initialdata=np.stack(cols, axis=1))

    def get_impulses(self, x):
        """x: (R*L,) or (R*L, C) capture (or an Audio). Returns an array (K, L, C): [k, :, c] is the response from
        output k to capture channel c. Same checks as get_impulse (mls.py:322-323, 298)."""
        if isinstance(x, Audio):
            x = x.samples
        assert isinstance(x, np.ndarray)
        assert len(x) > self.L, "The first sequence will be thrown away"
        if len(x) % self.L:
            raise ValueError("number of frames %d is not a multiple of L=%d" % (len(x), self.L))
        x2, code, n_rows, n_ch, rs, cs = Plan._as_capture(x)
        span = ((n_rows - 1) * rs + (n_ch - 1) * cs + 1) * x2.itemsize
        d_x = DeviceBuffer(span, self.device)
        out = np.empty((len(self._plans), self.L, n_ch), dtype=np.float64)
        es = self._plans[0].np_dtype().itemsize
        d_h = DeviceBuffer(self.L * n_ch * es, self.device)
        try:
            _L.check(_L.lib().zmls_dev_upload(self.device, d_x.ptr, x2.ctypes.data, span))
            for k, p in enumerate(self._plans):
                p.reserve(n_ch, staging=True)
                p.impulse_device(d_x.ptr, code, n_rows, n_ch, rs, cs, d_h.ptr, n_ch, 1)
                sync(self.device)
                out[k] = d_h.download((self.L, n_ch), p.np_dtype)
        finally:
            d_x.free()
            d_h.free()
        return out

    def close(self):
        for p in self._plans:
            p.close()
