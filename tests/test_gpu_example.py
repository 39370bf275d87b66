"""examples/ex_mls.py end to end on the GPU against the same chain through the oracle (the scenario of zignal's
examples/ex_mls.py: emphasised MLS -> system -> de-emphasis -> averaged periods -> impulse response -> spectrum)."""
import os
import sys

import numpy as np
import pytest
import scipy.signal

from gpu_common import rel_err
from oracle import c_oracle, mls_oracle
from zignal_amd.filters.biquads import RBJ
from zignal_amd.measure.mls import TAPS

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,fs,repeats", [(16, 48000, 5), (13, 96000, 4)])
def test_example_chain_matches_the_oracle_chain(N, fs, repeats):
    import ex_mls
    mls, z, ir, (freqs, mag) = ex_mls.measure(N, fs, repeats, device=0)
    taps = TAPS[N][0]
    L = (1 << N) - 1
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=fs).get_coefficients()
    seq = np.asarray(c_oracle.sequence(N, taps), dtype=np.float64).reshape(-1, 1)
    stim = scipy.signal.lfilter(B, A, np.tile(seq, (repeats, 1)), axis=0)          # MLS.apply_emphasis (mls.py:393-399)
    assert rel_err(mls.samples, stim) <= 1e-12
    y = ex_mls.system_under_test(stim, fs)
    z_want = scipy.signal.lfilter(A, B, y.samples, axis=0)                         # apply_deemphasis (mls.py:401-407)
    assert rel_err(z, z_want) <= 1e-11         # the stimulus fed to the system differs by 1e-16 already
    h_want = mls_oracle.get_impulse(N, taps, z_want, ref=seq)
    assert ir.samples.shape == (L, 1) and ir.fs == fs
    assert rel_err(ir.samples, h_want) <= 1e-11
    assert rel_err(ir.samples, mls_oracle.get_impulse(N, taps, z, ref=seq)) <= 1e-12     # the hot path on identical input
    if ex_mls.DELAY < L:
        assert int(np.argmax(np.abs(ir.samples))) == ex_mls.DELAY
    f_want, m_want = mls_oracle.audio_fft(ir.samples, fs, window='rectangular', normalise=False)
    assert np.array_equal(freqs, f_want)
    lin, lin_want = 10 ** (mag / 20), 10 ** (m_want / 20)
    assert np.max(np.abs(lin - lin_want)) <= 1e-12 * np.max(lin_want)


@pytest.mark.gpu
def test_plain_c_consumer_of_the_abi():
    """examples/c_api.c (C99, include/zmls.h only) built beforehand by build(): loop-back measurement through the host entry
    points, unit impulse within 1e-12, the reference's too-short-capture assertion as an error code."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "_build", "c_api")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        exe = g.build_c_example()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_api ok" in r.stdout, r.stdout + r.stderr
    assert "one period only -> -1" in r.stdout
