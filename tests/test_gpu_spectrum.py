"""GPU parity for the magnitude response: zmls_spectrum_* / Audio.fft against spectra
captured from the reference's Audio.fft (tests/golden/fft.npz) and against the oracle's restatement on larger sizes.
Floating point: linear magnitudes within 1e-12 x the largest bin (dB values of near-empty bins are ill-conditioned,
so dB is compared where the bin holds at least 1e-6 of the peak: there 1e-12 linear is < 1e-5 dB)."""
import json
import os

import numpy as np
import pytest

import inputs
from gpu_common import G
from oracle import mls_oracle
from zignal_amd.audio import Audio
from zignal_amd.device import Spectrum
from zignal_amd.measure.mls import MLS_simple

pytestmark = pytest.mark.gpu


def lin(db):
    return np.power(10.0, db / 20.0)


def check(mag, want):
    assert mag.shape == want.shape
    peak = np.max(lin(want))
    assert np.max(np.abs(lin(mag) - lin(want))) <= 1e-12 * peak
    strong = lin(want) >= 1e-6 * peak
    assert np.max(np.abs(mag[strong] - want[strong])) <= 1e-5


def test_audio_fft_against_reference_spectra():
    meta = json.load(open(os.path.join(G, "fft.json")))
    arrays = np.load(os.path.join(G, "fft.npz"))
    for c in meta["cases"]:
        rng = np.random.default_rng(c["seed"])
        x = rng.standard_normal((c["n"], c["ch"])) * np.exp(-np.arange(c["n"]) / (c["n"] / 8.0))[:, None]
        assert inputs.digest(x) == c["x_sha256"]
        frq, mag = Audio(fs=c["fs"], initialdata=x).fft(window=c["window"], normalise=c["normalise"], device=0)
        assert np.array_equal(frq, arrays[c["id"] + "_frq"])
        check(mag, arrays[c["id"] + "_mag"])


@pytest.mark.parametrize("N,ch", [(16, 1), (17, 2), (19, 1), (20, 1), (18, 5), (16, 4), (14, 3), (20, 2)])   # odd sizes with >= 2 channels: two channels per complex transform
def test_mls_sized_responses_against_the_oracle(N, ch):
    """L = 2^N - 1 points (odd, composite or Mersenne prime: N = 17, 19 get the size bump), rectangular window,
    not normalised: exactly what MLS_simple.plot_fft asks for (mls.py:466)."""
    L = (1 << N) - 1
    rng = np.random.default_rng(N)
    h = rng.standard_normal((L, ch)) * np.exp(-np.arange(L) / 3000.0)[:, None]
    sp = Spectrum(L, ch, device=0)
    assert sp.fftsize == L + (1 if N in (13, 17, 19, 31) else 0) and sp.n_bins == sp.fftsize // 2
    _, want = mls_oracle.audio_fft(h, 96000, "rectangular", False)
    check(sp.magnitude_db_host(h, None, False), want)
    _, want = mls_oracle.audio_fft(h, 96000, "hann", True)
    import scipy.signal.windows
    check(sp.magnitude_db_host(h, scipy.signal.windows.get_window("hann", Nx=L), True), want)
    sp.close()


def test_mls_simple_response_spectrum_is_the_emphasis_free_flat_line():
    ms = MLS_simple(N=12, fs=48000, repeats=3, device=0)
    k = ms.get_impulse(ms.samples)                # loop-back: a unit impulse (minus 1/L)
    frq, mag = k.fft(window='rectangular', normalise=False, device=0)
    assert frq.shape == (mag.shape[0],) and mag.shape == ((ms._mls.L) // 2, 1)
    assert np.max(np.abs(mag[1:])) < 1e-2         # flat 0 dB away from DC


def test_audio_fft_keeps_the_plan_of_a_shape_between_calls():
    """Audio.fft (audio.py:678-704) used to build and destroy a transform plan per call; the few most recent shapes are kept"""
    from zignal_amd import device as dev
    L = (1 << 12) - 1
    x = np.random.default_rng(3).standard_normal((L, 2))
    a = Audio(fs=48000, initialdata=x)
    _, m1 = a.fft(window='hann', device=0)
    sp = dev._spectra[(L, 2, 0)]
    _, m2 = a.fft(window='hann', device=0)
    assert dev._spectra[(L, 2, 0)] is sp and np.array_equal(m1, m2)
    for n in range(5):                                # more shapes than are kept: the oldest is closed and dropped
        Audio(fs=48000, initialdata=x[:100 + n]).fft(device=0)
    assert len(dev._spectra) == dev._SPECTRA_KEPT and (L, 2, 0) not in dev._spectra and sp._h is None
    _, m3 = a.fft(window='hann', device=0)            # ... and comes back on demand
    assert np.array_equal(m1, m3)


def test_a_quiet_channel_beside_a_loud_one_keeps_its_accuracy():
    """two channels share one complex transform at odd sizes; they are level-matched by a power of two first, so that the
    rounding error of the loud one (1e-16 x ITS peak) does not land in the quiet one's spectrum"""
    L = (1 << 14) - 1
    rng = np.random.default_rng(14)
    h = rng.standard_normal((L, 3)) * np.exp(-np.arange(L) / 500.0)[:, None]
    h[:, 1] *= 1e-10                                  # 200 dB below its partner
    h[:, 2] *= 1e7                                    # alone in its transform
    sp = Spectrum(L, 3, device=0)
    for win, norm in ((None, False), ("hann", True)):
        import scipy.signal.windows
        w = None if win is None else scipy.signal.windows.get_window(win, Nx=L)
        mag = sp.magnitude_db_host(h, w, norm)
        for c in range(3):
            _, want = mls_oracle.audio_fft(h[:, c:c + 1], 96000, win or "rectangular", norm)
            check(mag[:, c:c + 1], want)              # 1e-12 x the channel's OWN largest bin
    sp.close()


@pytest.mark.parametrize("n,ch", [(2, 1), (3, 2), (5, 1), (17, 3), (64, 2), (100, 2), (1000, 5), (4096, 2), (4097, 1), (8191, 2), (48000, 3),
                                  (1 << 15, 4), (1 << 16, 1), (100003, 2), ((1 << 19) - 1, 2), (1 << 20, 2), ((1 << 21) - 1, 2), ((1 << 22) - 1, 1),
                                  (3000000, 2)])
def test_any_length_against_numpy(n, ch):
    """Audio.fft works on whatever length the Audio holds (audio.py:680): powers of two run as forward passes whose output
    order the dB kernel gathers from (one tile, one strided + one contiguous pass, two strided passes at 2^23 points), every
    other length as the fused Bluestein chain - tiny ones with several transforms per tile, 2^22 - 1 with three passes each way."""
    rng = np.random.default_rng(n)
    h = rng.standard_normal((n, ch)) * np.exp(-np.arange(n) / max(4.0, n / 6.0))[:, None]
    sp = Spectrum(n, ch, device=0)
    assert sp.fftsize == n + (1 if n in (8191, (1 << 19) - 1) else 0)
    for win, norm in ((None, False), ("hamming", True)):
        import scipy.signal.windows
        w = None if win is None else scipy.signal.windows.get_window(win, Nx=n)
        _, want = mls_oracle.audio_fft(h, 48000, win or "rectangular", norm)
        check(sp.magnitude_db_host(h, w, norm), want)
    sp.close()


def test_time_contiguous_responses_take_the_element_kernels():
    """responses stacked channel by channel (ch_stride = L, row_stride = 1) go through the element-wise pack / dB kernels"""
    from zignal_amd.device import DeviceBuffer
    n, ch = (1 << 14) - 1, 3
    rng = np.random.default_rng(5)
    h = rng.standard_normal((n, ch)) * np.exp(-np.arange(n) / 900.0)[:, None]
    sp = Spectrum(n, ch, device=0)
    d_h = DeviceBuffer(n * ch * 8).upload(np.ascontiguousarray(h.T))
    d_m = DeviceBuffer(sp.n_bins * ch * 8)
    sp.magnitude_db_device(d_h.ptr, 1, n, None, False, d_m.ptr)
    got = d_m.download((sp.n_bins, ch), np.float64)
    _, want = mls_oracle.audio_fft(h, 48000, "rectangular", False)
    check(got, want)
    sp.close(); d_h.free(); d_m.free()


def test_a_response_of_fifty_million_points():
    """three levels each way (n = 3 x 2^24, a length numpy transforms quickly; M = 2^27 as 8 + 7 + 12 bits), offsets beyond 2^31 bytes
    inside one transform"""
    n = 3 << 24
    rng = np.random.default_rng(26)
    h = (rng.standard_normal(n) * np.exp(-np.arange(n) / 2.0e6)).reshape(-1, 1)
    sp = Spectrum(n, 1, device=0)
    mag = sp.magnitude_db_host(h, None, False)
    sp.close()
    want = 20.0 * np.log10(np.abs(np.fft.fft(h[:, 0])[:n // 2]))
    assert mag.shape == (n // 2, 1)
    check(mag[::4099], want[::4099].reshape(-1, 1))
    peak = np.max(lin(want))
    assert np.max(np.abs(lin(mag[:, 0]) - lin(want))) <= 1e-12 * peak


def test_more_pairs_than_go_through_the_passes_at_once():
    """Bluestein runs 256 MiB of transforms at a time (N=16: 128 pairs of 2^17 points): 300 channels are two chunks, the
    second one short"""
    n, ch = (1 << 16) - 1, 300
    rng = np.random.default_rng(300)
    h = rng.standard_normal((n, ch)) * np.exp(-np.arange(n) / 4000.0)[:, None] * (1.0 + np.arange(ch))[None, :]
    sp = Spectrum(n, ch, device=0)
    mag = sp.magnitude_db_host(h, None, True)
    sp.close()
    for c in (0, 1, 127, 255, 256, 257, 298, 299):
        _, want = mls_oracle.audio_fft(h[:, c:c + 1], 48000, "rectangular", True)
        check(mag[:, c:c + 1], want)
