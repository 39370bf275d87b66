"""GPU parity for the emphasis / de-emphasis filter: zmls_iir_* against
scipy.signal.lfilter, the very routine the reference calls (zignal/filters/linearfilter.py:62-63), on the same
inputs. Tolerance 1e-12 x peak (float64 arithmetic on both sides; the block-parallel start states differ from the
sequential ones by rounding only)."""
import numpy as np
import pytest
import scipy.signal

from gpu_common import rel_err
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, IirFilter, sync
from zignal_amd.filters.biquads import RBJ
from zignal_amd.measure.mls import MLS, TAPS, MLS_simple

pytestmark = pytest.mark.gpu
TOL = 1e-12


def stable_filter(order, rng):
    """random minimum-phase B and A of the given order (roots inside the unit circle)"""
    def poly():
        r = rng.uniform(0.2, 0.97, size=order) * np.exp(1j * rng.uniform(0, np.pi, size=order))
        roots = []
        for k in range(0, order - 1, 2):
            roots += [r[k], np.conj(r[k])]
        if order % 2:
            roots.append(rng.uniform(-0.95, 0.95))
        return np.real(np.poly(roots)) if roots else np.array([1.0])
    return rng.uniform(0.5, 2.0) * poly(), rng.uniform(0.5, 2.0) * poly()


@pytest.mark.parametrize("order", range(0, 9))
def test_every_order_against_lfilter_single_channel(order):
    rng = np.random.default_rng(100 + order)
    B, A = stable_filter(order, rng)
    f = IirFilter(B, A, device=0)
    for n in (1, 31, 32, 33, 8191, 8192, 8193, 100003):          # around the sub-block and chunk sizes
        x = rng.standard_normal(n)
        want = scipy.signal.lfilter(B, A, x, axis=0)
        assert rel_err(f.filter_host(x), want) <= TOL
        assert f.filter_host(x.reshape(-1, 1)).shape == (n, 1)
    f.close()


@pytest.mark.parametrize("shape,dtype", [((70000, 3), np.float64), ((20011, 16), np.float64), ((9000, 37), np.float32),
                                         ((50000, 5), np.int16), ((4097, 1), np.int32),
                                         # the chunks' start states come from the second-level scan: 586 chunks = 3 groups of
                                         # 256 frame-major, 37 chunks = 1 group time-contiguous; then 18 and 2 groups
                                         ((300007, 6), np.float64), ((2300000, 2), np.float32)])
def test_frames_by_channels_and_dtypes(shape, dtype):
    rng = np.random.default_rng(shape[0] + shape[1])
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()
    if np.issubdtype(dtype, np.integer):
        x = rng.integers(-30000, 30000, size=shape).astype(dtype)
    else:
        x = rng.standard_normal(shape).astype(dtype)
    for b, a in ((B, A), (A, B)):                                # emphasis and its inverse (mls.py:365-366)
        f = IirFilter(b, a, device=0)
        want = scipy.signal.lfilter(b, a, x.astype(np.float64), axis=0)
        got = f.filter_host(x)
        assert got.dtype == np.float64 and got.shape == x.shape
        assert rel_err(got, want) <= TOL
        xt = np.ascontiguousarray(x.T).T                          # same values, time-contiguous memory
        assert rel_err(f.filter_host(xt), want) <= TOL
        f.close()


def test_device_entry_point_in_place_and_workspace_error():
    rng = np.random.default_rng(3)
    n, C = 40000, 8
    x = rng.standard_normal((n, C))
    B, A = stable_filter(2, rng)
    f = IirFilter(B, A, device=0)
    d = DeviceBuffer(x.nbytes).upload(x)
    with pytest.raises(_lib.ZmlsError) as ei:
        f.filter_device(d.ptr, _lib.ZMLS_F64, n, C, C, 1, d.ptr, C, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    f.reserve(n, C)
    f.filter_device(d.ptr, _lib.ZMLS_F64, n, C, C, 1, d.ptr, C, 1)          # in place
    sync(0)
    assert rel_err(d.download((n, C), np.float64), scipy.signal.lfilter(B, A, x, axis=0)) <= TOL
    f.close()
    with pytest.raises(_lib.ZmlsError) as ei:
        IirFilter(np.ones(10), np.ones(3), device=0)
    assert ei.value.code == _lib.E_UNSUPPORTED


def test_long_capture_like_the_headline_stimulus():
    """(R*L, 1) at N=20, R=3: 3.1 M samples, 384 chunks; de-emphasis of an emphasised MLS gives the MLS back."""
    N, R = 20, 3
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()
    m = MLS(N=N, taps=TAPS[N][0], fs=96000, repeats=R, B=B, A=A, device=0)
    s0 = m.samples.copy()
    m.apply_emphasis()
    assert rel_err(m.samples, scipy.signal.lfilter(B, A, s0, axis=0)) <= TOL
    back = m.apply_deemphasis(m.samples)
    assert np.max(np.abs(back - s0)) <= 1e-9
    k = m.get_impulse(back)
    assert abs(k.samples[0, 0] - 1.0) <= 1e-9


def test_mls_simple_fused_deemphasis_and_impulse_matches_the_two_steps():
    ms = MLS_simple(N=14, fs=48000, repeats=4, device=0)
    rng = np.random.default_rng(9)
    y = ms.samples + 1e-3 * rng.standard_normal(ms.samples.shape)
    k = ms.get_impulse(y)
    mls = ms._mls
    two = mls.get_impulse(mls.apply_deemphasis(y))
    assert k.samples.shape == two.samples.shape == (mls.L, 1)
    assert rel_err(k.samples, two.samples) <= 1e-14
    want = scipy.signal.lfilter(*mls._filter_deemphasis.get_coefficients(), y, axis=0)
    assert rel_err(mls.apply_deemphasis(y), want) <= TOL
    assert abs(k.samples[0, 0] - 1.0) < 1e-3
