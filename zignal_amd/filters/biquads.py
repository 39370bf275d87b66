"""RBJ "Audio EQ cookbook" biquads, normalised to a0 = 1 — the designer MLS users take their
emphasis filter from (zignal/filters/biquads.py:132-247; `MLS_simple` builds a -10 dB high shelf
at 100 Hz, zignal/measure/mls.py:433). Closed-form scalar math, host only."""
import numpy as np

from .linearfilter import Filter

_TYPES = ('lowpass', 'highpass', 'bandpass1', 'bandpass2', 'notch', 'allpass', 'peak', 'lowshelf', 'highshelf')


class RBJ(Filter):
    class Types(object):
        lowpass, highpass, bandpass1, bandpass2 = 'lowpass', 'highpass', 'bandpass1', 'bandpass2'
        notch, allpass, peak, lowshelf, highshelf = 'notch', 'allpass', 'peak', 'lowshelf', 'highshelf'

    def __init__(self, filtertype=None, gaindb=0, f0=997, Q=0.707, fs=96000):
        Filter.__init__(self, B=(1, 0, 0), A=(1, 0, 0), fs=fs)
        self._filtertype, self._gaindb, self._f0, self._Q = filtertype, gaindb, f0, Q
        self.calculate_coefficients(filtertype=filtertype, gaindb=gaindb, f0=f0, Q=Q)

    def calculate_coefficients(self, filtertype=None, gaindb=None, f0=None, Q=None):
        if filtertype not in _TYPES:
            raise NotImplementedError("Valid types are: %s" % list(_TYPES))
        amp = 10 ** (gaindb / 40)
        w0 = 2 * np.pi * f0 / self.fs
        cw, sw = np.cos(w0), np.sin(w0)
        alpha = sw / (2 * Q)
        # the seven second-order sections that share the denominator 1 + alpha, -2cos, 1 - alpha
        plain_den = (1 + alpha, -2 * cw, 1 - alpha)
        if filtertype == 'lowpass':
            num, den = ((1 - cw) / 2, 1 - cw, (1 - cw) / 2), plain_den
        elif filtertype == 'highpass':
            num, den = ((1 + cw) / 2, -(1 + cw), (1 + cw) / 2), plain_den
        elif filtertype == 'bandpass1':
            num, den = (sw / 2, 0, -sw / 2), plain_den
        elif filtertype == 'bandpass2':
            num, den = (alpha, 0, -alpha), plain_den
        elif filtertype == 'notch':
            num, den = (1, -2 * cw, 1), plain_den
        elif filtertype == 'allpass':
            num, den = (1 - alpha, -2 * cw, 1 + alpha), plain_den
        elif filtertype == 'peak':
            num = (1 + alpha * amp, -2 * cw, 1 - alpha * amp)
            den = (1 + alpha / amp, -2 * cw, 1 - alpha / amp)
        else:
            beta = 2 * np.sqrt(amp) * alpha
            sgn = 1 if filtertype == 'lowshelf' else -1     # the two shelves mirror each other in cos(w0)
            num = (amp * ((amp + 1) - sgn * (amp - 1) * cw + beta),
                   sgn * 2 * amp * ((amp - 1) - sgn * (amp + 1) * cw),
                   amp * ((amp + 1) - sgn * (amp - 1) * cw - beta))
            den = ((amp + 1) + sgn * (amp - 1) * cw + beta,
                   -sgn * 2 * ((amp - 1) + sgn * (amp + 1) * cw),
                   (amp + 1) + sgn * (amp - 1) * cw - beta)
        self.set_coefficients(B=num, A=den)
        self.normalise()
