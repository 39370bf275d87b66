"""Several MLS stimuli at once: one output channel per tap set, every capture channel paired with every
stimulus at extraction — SURVEY.md §8(f) row 3.

The reference describes this and leaves it to the caller (zignal/measure/mls.py:338-343: "creating multiple
instances ... constructing a new Audio instance with all channels appended. To extract the impulse response we
then need to keep track of which channel is paired with which MLS since they should preferably be using
different taps to minimise cross talk"). Here it is a front end over one device plan per tap set and ONE device
call per capture (zmls_impulse_sources_device, csrc/sources.hip): the capture is uploaded once, its periods are added
up once whatever the number of stimuli (the average does not depend on the taps), every plan then gathers, transforms
and scatters that one period of sums on a workspace all plans share, so the (output k -> input c) response is
`impulses[k][:, c]`. Each pair equals what `_MLS_base(N, taps_k).get_impulse(x[:, c])` returns
(tests/golden/multi_source.*: the reference run pair by pair).
"""
import numpy as np

from ..audio import Audio
from ..device import IirFilter, Plan, Sources
from ..filters.linearfilter import Filter
from .mlstaps import TAPS


class MultiMLS(object):
    def __init__(self, N=None, taps=None, fs=96000, repeats=2, B=(1, 0, 0), A=(1, 0, 0), dtype='f64', device=None):
        """taps: a sequence of tap sets, one per output channel (all different), or an int K to take the first K
        sets of TAPS[N]. B, A: emphasis filter shared by all stimuli, minimum phase as for MLS (mls.py:351-369)."""
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
        self._filter_emphasis = Filter(B=B, A=A, fs=fs)
        self._filter_deemphasis = Filter(B=A, A=B, fs=fs)       # inverse filter
        assert self._filter_emphasis.is_minimum_phase(), \
            "The emphasis filter must be minimum phase, i.e. possible to invert"
        self._plans = [Plan(N, t, dtype=dtype, device=device) for t in taps]
        self._sources = Sources(self._plans)
        self.device = self._plans[0].device
        self._iirs = {}

    def __len__(self):
        return len(self._plans)

    def __repr__(self):
        B, A = self._filter_emphasis.get_coefficients()
        return 'MultiMLS(N=%i, taps=%s, fs=%r, repeats=%i, B=%s, A=%s)' % (self.N, self.taps, self.fs, self.repeats, tuple(B), tuple(A))

    def _iir(self, which):
        if which not in self._iirs:
            B, A = (self._filter_emphasis if which == 'e' else self._filter_deemphasis).get_coefficients()
            if max(len(B), len(A)) - 1 > IirFilter.MAX_ORDER:
                raise NotImplementedError("emphasis filters beyond order %d are not built for the device" % IirFilter.MAX_ORDER)
            self._iirs[which] = IirFilter(B, A, device=self.device)
        return self._iirs[which]

    def get_stimulus(self, emphasis=False):
        """Audio with one channel per tap set: (repeats*L, K) float64, channel k = get_full_sequence of tap set k,
        appended channel by channel as the reference's recipe does (audio.py:376-421). emphasis=True: every channel
        filtered with the emphasis filter (MLS.apply_emphasis, mls.py:393-399), one device call for all of them."""
        stim = Audio(fs=self.fs, initialdata=self._plans[0].sequence(self.repeats))
        stim.append(*[Audio(fs=self.fs, initialdata=p.sequence(self.repeats)) for p in self._plans[1:]])
        if emphasis:
            stim.samples = self._iir('e').filter_host(stim.samples)
        return stim

    def get_impulses(self, x, deemphasis=False, outputs=None):
        """x: (R*L,) or (R*L, C) capture (or an Audio). Returns an array (K, L, C): [k, :, c] is the response from
        output k to capture channel c. Same checks as get_impulse (mls.py:322-323, 298). deemphasis=True: the capture
        is filtered with the inverse of the emphasis filter first (MLS.apply_deemphasis, mls.py:401-407), inside the
        same device call. outputs: indices of the stimuli wanted (default all)."""
        if isinstance(x, Audio):
            x = x.samples
        assert isinstance(x, np.ndarray)
        assert len(x) > self.L, "The first sequence will be thrown away"
        if len(x) % self.L:
            raise ValueError("number of frames %d is not a multiple of L=%d" % (len(x), self.L))
        group = self._sources if outputs is None else Sources([self._plans[k] for k in outputs])
        return group.impulse_host(x, iir=self._iir('d') if deemphasis else None).astype(np.float64, copy=False)

    def get_impulse(self, x, pairs=None, deemphasis=False):
        """The pairing the reference asks its user to keep track of: responses of (capture channel c, output k) pairs,
        0-based, as an Audio with one channel per pair in the order given (default: every pair, capture channel by
        capture channel)."""
        data = x.samples if isinstance(x, Audio) else np.asarray(x)
        n_in = 1 if data.ndim == 1 else data.shape[1]
        if pairs is None:
            pairs = [(c, k) for c in range(n_in) for k in range(len(self._plans))]
        for c, k in pairs:
            assert 0 <= c < n_in and 0 <= k < len(self._plans), "pair (%d, %d) is out of range" % (c, k)
        used = sorted(set(k for _, k in pairs))
        h = self.get_impulses(data, deemphasis=deemphasis, outputs=used)
        return Audio(fs=self.fs, initialdata=np.stack([h[used.index(k), :, c] for c, k in pairs], axis=1))

    def close(self):
        for f in self._iirs.values():
            f.close()
        self._iirs = {}
        for p in self._plans:
            p.close()
