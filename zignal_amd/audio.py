"""Minimal stand-in for zignal's `Audio` container — only what the MLS path touches.

The hot path stores its stimulus in `Audio.samples` and hands the impulse response back as an
`Audio` (zignal/measure/mls.py:361, 413). This shim reproduces the constructor contract of
zignal/audio.py:25-104 for the `initialdata` case (copy, 1-D -> column, shape checks, fs /
nofsamples / ch / duration bookkeeping) and the few read-only helpers `__str__` needs
(audio.py:106-125, 463-531). Everything else in zignal's Audio (wav I/O, resampling, fades,
plotting, ...) is outside the scope of this library.
"""
import logging

import numpy as np


class Audio(object):
    def __init__(self, channels=0, fs=96000, nofsamples=0, duration=None, initialdata=None, dtype=np.float64):
        self._logger = logging.getLogger(__name__)
        assert fs > 0, "sample rate cannot be zero or negative"
        self.fs = fs
        self._comment = ''
        if initialdata is None:
            assert isinstance(channels, int)
            assert not (nofsamples != 0 and duration is not None), "choose either samples or duration"
            self.ch = channels
            if duration is not None:
                self.nofsamples, self.duration = int(duration * fs), duration
            else:
                self.nofsamples, self.duration = nofsamples, nofsamples / fs
            self.samples = np.zeros((self.nofsamples, self.ch), dtype=dtype)
            return
        assert isinstance(initialdata, np.ndarray), 'Only numpy arrays are allowed as initial data'
        assert channels == 0, "parameter 'channels' is redundant if initial data is specified"
        assert nofsamples == 0, "parameter 'nofsamples' is redundant if initial data is specified"
        assert duration is None, "parameter 'duration' is redundant if initial data is specified"
        data = initialdata.copy()                     # callers keep ownership of what they passed in
        if data.ndim == 1:
            data = data[:, np.newaxis]
        assert data.ndim == 2, 'shape must be (Nsamples, Nchannels)'
        self.samples = data
        self.nofsamples, self.ch = data.shape
        assert self.nofsamples > self.ch, 'shape must be (Nsamples, Nchannels)'
        self.duration = self.nofsamples / self.fs

    def __len__(self):
        return self.nofsamples

    def copy(self):
        return Audio(fs=self.fs, initialdata=self.samples)

    def append(self, *args):
        """Add the channels of other Audio instances to the right of the current ones; the shorter side is padded with
        zeros (audio.py:376-421). This is how the reference builds a multichannel stimulus out of several MLS
        (mls.py:338-343)."""
        for other in args:
            assert isinstance(other, Audio), "only Audio() instances can be used"
            assert self.fs == other.fs, "Sample rates must match (%s != %s)" % (self.fs, other.fs)
            assert self.samples.dtype == other.samples.dtype, \
                "Data types must match (%s != %s)" % (self.samples.dtype, other.samples.dtype)
            rows = max(self.nofsamples, other.nofsamples)
            both = np.zeros((rows, self.ch + other.ch), dtype=self.samples.dtype)
            both[:self.nofsamples, :self.ch] = self.samples
            both[:other.nofsamples, self.ch:] = other.samples
            self.samples = both
            self.nofsamples, self.ch = both.shape
            self.duration = self.nofsamples / self.fs

    # -- sample formats (host side; the MLS path can take the integers directly: get_impulse_pcm) ----------
    def convert_to_float(self, targetbits=64):
        """Integers -> floats in [-1, 1): samples / 2^(sourcebits-1), in place (audio.py:565-589)."""
        assert targetbits in (32, 64)
        target = np.float32 if targetbits == 32 else np.float64
        if np.issubdtype(self.samples.dtype, np.integer):
            full_scale = 2 ** (8 * self.samples.itemsize - 1)
            self.samples = np.array(self.samples / full_scale, dtype=target)
        else:
            self.samples = np.array(self.samples, dtype=target)

    def convert_to_integer(self, targetbits=16):
        """Floats in [-1, 1] -> signed integers: samples * (2^(targetbits-1) - 1), truncated, in place
        (audio.py:534-563)."""
        assert targetbits in (8, 16, 32, 64)
        if not np.issubdtype(self.samples.dtype, np.floating):
            raise NotImplementedError("TODO: implement scale int->int")
        self.samples = np.array(self.samples * (2 ** (targetbits - 1) - 1), dtype='int%d' % targetbits)

    # -- per-channel statistics shown by __str__ -------------------------------------------------
    def _wide(self):
        s = self.samples
        return s if np.issubdtype(s.dtype, np.floating) else s.astype(np.int64)

    def peak(self):
        """(signed peak value, its row index) per channel"""
        if len(self.samples) == 0:
            return np.full(self.ch, np.nan), np.zeros(self.ch, dtype=np.int64)
        idx = np.abs(self._wide()).argmax(axis=0)
        return np.array([self.samples[r, c] for c, r in enumerate(idx)]), idx

    def rms(self):
        if len(self.samples) == 0:
            return np.full(self.ch, np.nan)
        if (self.samples == 0).all():
            return np.zeros(self.ch)
        return np.sqrt(np.mean(np.power(self._wide(), 2), axis=0))

    def crest_factor(self):
        if len(self.samples) == 0:
            return np.full(self.ch, np.nan)
        with np.errstate(invalid='ignore', divide='ignore'):
            return np.abs(self.samples).max(axis=0) / self.rms()

    def delay(self, n, channel=1):
        """Delay one channel (1-indexed) by n samples, zeros shifted in (audio.py:328-331)."""
        col = self.samples[:, channel - 1]
        self.samples[:, channel - 1] = np.concatenate((np.zeros(n, dtype=col.dtype), col))[:len(col)]

    def __str__(self):
        rows = [
            '=======================================',
            'classname        : %s' % self.__class__.__name__,
            'sample rate      : %.1f [Hz]' % self.fs,
            'channels         : %i' % self.ch,
            'duration         : %.3f [s]' % self.duration,
            'datatype         : %s' % self.samples.dtype,
            'samples per ch   : %i' % self.nofsamples,
            'data size        : %.3f [Mb]' % (self.samples.nbytes / (1024 * 1024)),
            'has comment      : %s' % ('yes' if len(self._comment) != 0 else 'no'),
        ]
        if self.ch != 0:
            fmt = dict(precision=4, suppress_small=True)
            rows.append('peak             : %s' % np.array_str(self.peak()[0], **fmt))
            rows.append('RMS              : %s' % np.array_str(self.rms(), **fmt))
            rows.append('crestfactor      : %s' % np.array_str(self.crest_factor(), **fmt))
        rows.append('-----------------:---------------------')
        return "\n".join(rows) + "\n"

    def fft(self, window='hann', normalise=True, device=None):
        """FFT magnitude of all channels in dB, up to fs/2: (frq, mag) as zignal/audio.py:678-704. The transform
        runs on the GPU (zmls_spectrum_*: window, this library's DFT passes, dB); window values and the frequency axis are host work."""
        import scipy.signal.windows
        from .device import spectrum_for
        sp = spectrum_for(self.nofsamples, self.ch, device=device)     # the plan of this shape is kept between calls
        rect = window in ('rectangular', 'boxcar', 'box', 'ones', 'rect')
        win = None if rect else scipy.signal.windows.get_window(window, Nx=self.nofsamples)   # not fftsize! (audio.py:690)
        mag = sp.magnitude_db_host(self.samples, win, normalise)
        # np.fft.fftfreq(fftsize, 1 / fs)[:int(fftsize / 2)] (audio.py:699-701) without its upper half: fftfreq is integers times
        # 1.0 / (n * d), so these are the same doubles; kept with the shape's Spectrum, handed out as a copy
        key = float(self.fs)
        if getattr(sp, "_frq_fs", None) != key:
            sp._frq = np.arange(int(sp.fftsize / 2)) * (1.0 / (sp.fftsize * (1 / self.fs)))
            sp._frq_fs = key
        return sp._frq.copy(), mag

    def plot_fft(self, plotname=None, window='hann', normalise=True, **kwargs):
        """Magnitude response plot with the reference's defaults (audio.py:649-676: hann window, one labelled line per
        channel, nothing below 1 Hz, floor `ymin` = -160 dB). The spectrum comes from `fft` (GPU); the drawing is matplotlib's."""
        import matplotlib.pyplot as plt
        ymin = kwargs.get('ymin', -160)
        freq, mag = self.fft(window=window, normalise=normalise)
        fig = plt.figure()
        for ch in range(self.ch):
            plt.semilogx(freq, mag[:, ch], label='ch%2i' % (ch + 1))
        plt.xlim(left=1)
        plt.ylim(bottom=ymin)
        plt.xlabel('Frequency [Hz]')
        plt.ylabel('Magnitude [dB]')
        plt.legend(loc='best')
        plt.grid(True)
        if plotname is None:
            plt.show()
        else:
            plt.savefig(plotname)
            plt.close(fig)
