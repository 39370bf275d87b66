"""GPU parity at BASELINE.json's full single-GPU sizes (configs 2-5 and one GPU's share of config 3).

The batches are far too large to push through the oracle, so every case is held to
  * the oracle on a few channels pulled back from the device (bit-for-bit the same capture), and
  * a size-independent property of every sampled channel: the synthetic channel c is the stimulus through a
    known 16-tap FIR g_c delayed by d_c, and an MLS has a two-valued circular autocorrelation (L at lag 0, -1
    elsewhere), so without noise the response must be  h = g_c(1 + 1/L) - sum(g_c)/L  at lag d_c, and
    sum_m h[m] = -(1/L) sum_j avg[j]  for any capture.
Captures are generated on the device (zmls_synth_device), in zignal's (frames, channels) layout.
Tolerances: 1e-12 x peak for f64, 1e-5 x peak for the f32 mode (BASELINE.json north_star)."""
import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from oracle import c_oracle
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, download_channel, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu

SEED = 0x5A17


def expected_noise_free(p, c, seed=SEED):
    g, d = p.synth_system(c, seed)
    want = np.zeros(p.L)
    want[(d + np.arange(16)) % p.L] = g
    return want * (1.0 + 1.0 / p.L) - g.sum() / p.L


def run_batch(N, C, R, dtype, sigma):
    p = Plan(N, TAPS[N][0], dtype=dtype, device=0)
    es = 8 if dtype == "f64" else 4
    d_x, _ = p.alloc_capture(R * p.L, C, tries=2)
    d_h = DeviceBuffer(p.L * C * es)
    d_h.fill(0xFF)
    p.reserve(C)
    p.synth_device(d_x.ptr, p.dtype_code, R, C, C, 1, seed=SEED, sigma=sigma)
    p.impulse_device(d_x.ptr, p.dtype_code, R * p.L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    return p, d_x, d_h


def channel(p, d_buf, n_rows, C, c):
    return download_channel(0, d_buf.ptr, p.dtype_code, n_rows, C, 1, c).astype(np.float64)


def one_per_slab(C, slab=16):
    """one channel of every 16-channel block (the widest tile's slab in f64; two per slab in f32), at a position that
    walks through the slab, so that a defect confined to one slab - or to one lane group of every slab - shows"""
    return [s * slab + (5 * s + 3) % slab for s in range(C // slab)]


def check_channels(p, d_x, d_h, C, R, chans, tol, oracle_chans, sigma, capture_chans=None):
    """known-system check on every channel of `chans` (one download of the response each); the sum identity, which
    needs the capture too, on `capture_chans` (default: all of them); the oracle on `oracle_chans`"""
    ref = c_oracle.sequence(p.N, p.taps)
    capture_chans = set(chans if capture_chans is None else capture_chans) | set(oracle_chans)
    for c in sorted(set(chans) | capture_chans):
        h = channel(p, d_h, p.L, C, c)
        assert np.all(np.isfinite(h)), c
        if sigma == 0.0:
            assert np.max(np.abs(h - expected_noise_free(p, c))) <= tol, c
        else:
            # noise averages down by sqrt(periods * L): far below 1e-4 of the unit peak at these sizes
            assert np.max(np.abs(h - expected_noise_free(p, c))) <= 1e-4, c
        if c not in capture_chans:
            continue
        x = channel(p, d_x, R * p.L, C, c)
        avg = x[p.L:].reshape(R - 1, p.L).mean(axis=0)
        # sum of the response = -(1/L) sum of the averaged capture (autocorrelation sums to 1 per lag: L - (L-1))
        assert abs(h.sum() + avg.sum() / p.L) <= tol * max(1.0, np.abs(avg).sum() / p.L) * (10 if tol < 1e-9 else 1000), c
        if c in oracle_chans:
            assert rel_err(h, oracle_ir(p.N, p.taps, x.reshape(-1, 1), ref)) <= tol, c


def test_config2_n18_64_channels_f64():
    N, C, R = 18, 64, 2
    p, d_x, d_h = run_batch(N, C, R, "f64", 1e-3)
    assert p.info()["passes"] == 2
    check_channels(p, d_x, d_h, C, R, range(0, C, 7), 1e-12, {0, 21, 63}, 1e-3)
    p.close()


def test_config3_share_n20_512_channels_16_periods_f64():
    N, C, R = 20, 512, 17
    p, d_x, d_h = run_batch(N, C, R, "f64", 1e-3)
    # 32 slabs of 16 channels: one channel of each, plus the corners
    check_channels(p, d_x, d_h, C, R, one_per_slab(C) + [0, 15, 16, 511], 1e-12, {0, 511}, 1e-3, capture_chans=(0, 15, 16, 255, 511))
    d_x.free()
    d_h.free()
    p.close()


def test_config3_share_with_deemphasis_in_one_call_both_forms():
    """MLS_simple.get_impulse's chain (apply_deemphasis + get_impulse, mls.py:401-414, 454-458) on one GPU's share of config 3 with
    the reference's shelving filter (mls.py:433), through zmls_impulse_filtered_device in its two forms: the filter after the
    transform (the default at this size) and the one sweep in time order. Sampled channels: the two forms against each other, the
    known system (its response through the filter, circularly: the filter's memory is a hundredth of a period), and lfilter +
    oracle on the very capture for the two corner channels."""
    import scipy.signal
    from zignal_amd.device import IirFilter
    from zignal_amd.filters.biquads import RBJ
    N, C, R = 20, 512, 17
    p, d_x, d_h = run_batch(N, C, R, "f64", 1e-3)
    L = p.L
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()
    f = IirFilter(A, B, device=0)                                  # de-emphasis: the inverse of the emphasis filter (mls.py:365-366)
    f.reserve_fused(p, R * L, C)
    chans = one_per_slab(C) + [0, 511]
    got = {}
    for form in (1, 0):
        f.set_fused_form(form)
        d_h.fill(0xFF)
        p.impulse_filtered_device(f, d_x.ptr, p.dtype_code, R * L, C, C, 1, d_h.ptr, C, 1)
        sync(0)
        assert f.last_fused_form() == form
        got[form] = {c: channel(p, d_h, L, C, c) for c in chans}
    f.set_fused_form(-1)
    p.impulse_filtered_device(f, d_x.ptr, p.dtype_code, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    assert f.last_fused_form() == 1
    ref = c_oracle.sequence(N, p.taps)
    for c in chans:
        assert np.all(np.isfinite(got[1][c])), c
        assert rel_err(got[1][c], got[0][c]) <= 1e-13, c
        e = expected_noise_free(p, c)
        want = scipy.signal.lfilter(A, B, np.concatenate([e, e]))[L:]          # the periodic steady state = the circular filter
        assert np.max(np.abs(got[1][c] - want)) <= 1e-4, c
    for c in (0, 511):
        x = channel(p, d_x, R * L, C, c)
        y = scipy.signal.lfilter(A, B, x)
        assert rel_err(got[1][c], oracle_ir(N, p.taps, y.reshape(-1, 1), ref)) <= 1e-12, c
    f.close()
    d_x.free()
    d_h.free()
    p.close()


def test_config4_n24_256_channels_three_passes_f64():
    N, C, R = 24, 256, 2
    p, d_x, d_h = run_batch(N, C, R, "f64", 0.0)
    assert p.info()["passes"] == 3
    check_channels(p, d_x, d_h, C, R, one_per_slab(C) + [0, 255], 1e-12, {100}, 0.0, capture_chans=(0, 100, 255))
    d_x.free()
    d_h.free()
    p.close()


def test_config5_n20_1024_channels_f32_against_f64():
    N, C, R = 20, 1024, 2
    p64, d_x64, d_h64 = run_batch(N, C, R, "f64", 1e-3)
    chans = (0, 333, 1023)
    h64 = {c: channel(p64, d_h64, p64.L, C, c) for c in chans}
    check_channels(p64, d_x64, d_h64, C, R, one_per_slab(C) + list(chans), 1e-12, {333}, 1e-3, capture_chans=chans)
    d_x64.free()
    d_h64.free()
    p64.close()
    p32, d_x32, d_h32 = run_batch(N, C, R, "f32", 1e-3)
    for c in chans:
        # same synthetic capture rounded to float32, float32 arithmetic: the accuracy side of the trade-off
        assert rel_err(channel(p32, d_h32, p32.L, C, c), h64[c]) <= 1e-5
    check_channels(p32, d_x32, d_h32, C, R, one_per_slab(C) + list(chans), 1e-5, {1023}, 1e-3, capture_chans=chans)
    d_x32.free()
    d_h32.free()
    p32.close()
