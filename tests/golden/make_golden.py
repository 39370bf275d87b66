#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REAL reference
(/root/reference, zignal v0.6.0) on inputs this repository owns.

Runs only in the build container: it refuses to start without /root/reference
and nothing under tests/, bench.py or the package imports it. Fixtures are data
(inputs, outputs, digests); no reference source is copied.

Import recipe is the one recorded in SURVEY.md §8(c): the reference's
`audio.py` imports the (absent) `samplerate` package at module level although
the MLS path never calls it, so an empty module object is registered first.

    python tests/golden/make_golden.py            # everything (≈6 min)
    python tests/golden/make_golden.py --quick    # skip N >= 21 sequence KATs and N=24 IR
"""
import argparse
import json
import os
import sys
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

if not os.path.isdir(os.path.join(REF, "zignal")):
    sys.exit("make_golden.py needs the reference checkout at /root/reference")

sys.path.insert(0, HERE)
sys.path.insert(0, REF)
sys.modules.setdefault("samplerate", types.ModuleType("samplerate"))
os.environ.setdefault("MPLBACKEND", "Agg")
warnings.simplefilter("ignore")

import matplotlib  # noqa: E402

matplotlib.use("Agg")

import zignal  # noqa: E402  (the reference)
from zignal.measure.mls import MLS, MLS_simple, TAPS, _MLS_base, get_random_taps  # noqa: E402

import inputs  # noqa: E402  (ours)


def write_json(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


# --------------------------------------------------------------------------- taps
def dump_taps():
    table = {str(N): [list(t) for t in TAPS[N]] for N in sorted(TAPS)}
    txt = json.dumps(table, separators=(",", ":"), sort_keys=True)
    with open(os.path.join(HERE, "taps.json"), "w") as f:
        f.write(txt + "\n")
    # the package ships the same table as data (zignal_amd/measure/taps.json)
    with open(os.path.join(REPO, "zignal_amd", "measure", "taps.json"), "w") as f:
        f.write(txt + "\n")
    print("taps: %d orders, %d tap sets" % (len(table), sum(len(v) for v in table.values())))


# --------------------------------------------------------------------------- sequence KATs
def seq_kat(N, taps):
    m = _MLS_base(N, tuple(taps))
    s = m.get_full_sequence(repeats=1)
    assert s.shape == (m.L, 1) and s.dtype == np.float64
    packed = inputs.pack_bits_from_samples(s)
    bits = ((1.0 - s[:, 0]) / 2.0).astype(np.uint8)
    return {
        "N": N, "taps": list(taps), "ones": int(bits.sum()),
        "sha256": inputs.digest(packed),
        "first64": "".join(str(int(b)) for b in bits[:64]),
        "nbytes": int(packed.size),
    }


def dump_seq_kats(quick):
    out = []
    t0 = time.time()
    for N in sorted(TAPS):
        if N <= 12:
            sets = list(TAPS[N])
        elif N <= (20 if quick else 24):
            sets = [TAPS[N][0], TAPS[N][-1]]
            if N <= 16:
                sets.append(TAPS[N][len(TAPS[N]) // 2])
        else:
            continue
        for taps in sets:
            out.append(seq_kat(N, taps))
        print("seq KAT N=%d (%d sets) %.0fs" % (N, len(sets), time.time() - t0), flush=True)
    # generator_chunk continuity across the period boundary
    m = _MLS_base(5, TAPS[5][1])
    g = m.generator_chunk(20)
    chunks = [next(g) for _ in range(4)]
    assert chunks[0].shape == (20, 1) and chunks[0].dtype == np.int64
    cont = {"N": 5, "taps": list(TAPS[5][1]), "chunk": 20,
            "bits": [int(v) for c in chunks for v in c[:, 0]]}
    gs = m.generator_samples(7)
    sam = [next(gs) for _ in range(3)]
    cont["samples_chunk7"] = [float(v) for c in sam for v in c[:, 0]]
    write_json("seq_kat.json", {"kats": out, "generator_chunk": cont})


def dump_seq_prefix():
    """Orders beyond the full-period KATs (one N=32 period is ~100 minutes of the reference's per-bit loop): the first
    2^20 bits of the generator, straight from generator_chunk, for the first and last tap set of N = 25..32."""
    nbits = 1 << 20
    out = []
    t0 = time.time()
    for N in range(25, 33):
        for taps in (TAPS[N][0], TAPS[N][-1]):
            bits = next(_MLS_base(N, tuple(taps)).generator_chunk(nbits))[:, 0].astype(np.uint8)
            packed = np.packbits(bits, bitorder="little")
            out.append({"N": N, "taps": list(taps), "nbits": nbits, "ones": int(bits.sum()), "sha256": inputs.digest(packed),
                        "first64": "".join(str(int(b)) for b in bits[:64])})
        print("seq prefix N=%d %.0fs" % (N, time.time() - t0), flush=True)
    write_json("seq_prefix.json", {"prefixes": out})


def dump_seq_any():
    """Tap sets that are NOT in the reference's table but that its generator accepts all the same (it checks nothing,
    mls.py:64-68): not primitive, without N, a single tap, a tap listed twice, and taps N+1..2N, which the negative deque
    index at mls.py:107 maps onto 1..N. For each: the first L bits as get_full_sequence hands them out, and 3L+17 bits
    of generator_chunk, whose register keeps running (no period L to wrap at). Plus the taps its register cannot index."""
    sets = [(8, (8, 4)), (6, (6, 3)), (5, (5, 3, 2)), (8, (7, 3)), (10, (9, 4, 2)), (6, (12, 7)), (7, (14,)), (5, (5, 3, 3, 2)),
            (4, (4,)), (9, (3,)), (12, (12, 11, 10, 2)), (16, (16, 8)), (16, (15, 1)), (3, (1,)), (11, (22, 13)), (18, (18, 9))]
    out = []
    for N, taps in sets:
        m = _MLS_base(N, tuple(taps))
        s = m.get_full_sequence(repeats=2)
        assert s.shape == (2 * m.L, 1) and np.array_equal(s[:m.L], s[m.L:])
        first = ((1.0 - s[:m.L, 0]) / 2.0).astype(np.uint8)
        n_run = 3 * m.L + 17
        run = next(m.generator_chunk(n_run))[:, 0].astype(np.uint8)
        assert np.array_equal(run[:m.L], first)
        gb = m.generator_bit()
        assert [next(gb) for _ in range(min(n_run, 4000))] == [int(v) for v in run[:4000]]
        out.append({"N": N, "taps": list(taps), "ones": int(first.sum()), "sha256": inputs.digest(np.packbits(first, bitorder="little")),
                    "first64": "".join(str(int(b)) for b in first[:64]),
                    "run_bits": n_run, "run_ones": int(run.sum()), "run_sha256": inputs.digest(np.packbits(run, bitorder="little")),
                    "wraps_at_L": bool(np.array_equal(run[m.L:2 * m.L], first))})
        print("seq any N=%d taps=%s ones=%d wraps=%s" % (N, taps, out[-1]["ones"], out[-1]["wraps_at_L"]), flush=True)
    errors = []
    for N, taps in [(5, (0,)), (5, (5, 11)), (6, (6, -1)), (4, (4, 9))]:
        m = _MLS_base(N, tuple(taps))                    # the constructor accepts them
        try:
            next(m.generator_bit())
            raised = None
        except Exception as e:  # noqa: BLE001
            raised = type(e).__name__
        try:
            m.get_full_sequence(1)
            raised_full = None
        except Exception as e:  # noqa: BLE001
            raised_full = type(e).__name__
        errors.append({"N": N, "taps": list(taps), "generator_bit": raised, "get_full_sequence": raised_full})
        print("seq any error N=%d taps=%s -> %s / %s" % (N, taps, raised, raised_full), flush=True)
    write_json("seq_any.json", {"sets": out, "errors": errors})


# --------------------------------------------------------------------------- IR goldens
def ref_impulse(N, taps, x):
    m = _MLS_base(N, tuple(taps))
    avg = m.trim_and_avgerage(x)
    h = m.get_impulse(x)
    assert h.shape == (m.L, 1) and h.dtype == np.float64
    return h[:, 0].copy(), str(avg.dtype)


def dump_ir(quick):
    arrays = {}
    meta = []
    small = []
    for N in (3, 4, 5, 7, 8, 10, 11, 12):
        small.append((N, TAPS[N][0], "loopback", 2, 1, "col", "float64"))
        small.append((N, TAPS[N][-1], "noise", 4, 2, "flat", "float64"))
        small.append((N, TAPS[N][len(TAPS[N]) // 2], "system", 5, 3, "col", "float64"))
    small.append((10, TAPS[10][3], "system", 3, 4, "col", "float32"))
    small.append((10, TAPS[10][4], "system", 3, 5, "flat", "int16"))
    small.append((12, TAPS[12][7], "noise", 2, 6, "col", "float32"))
    small.append((13, TAPS[13][0], "system", 17, 7, "col", "float64"))
    small.append((14, TAPS[14][0], "system", 3, 8, "col", "float64"))
    for cid, (N, taps, kind, R, seed, shape, dtype) in enumerate(small):
        seq = _MLS_base(N, tuple(taps)).get_full_sequence(1)[:, 0]
        x = inputs.make_input(kind, seq, R, seed, shape, dtype)
        h, avg_dtype = ref_impulse(N, taps, x)
        key = "c%02d" % cid
        arrays[key + "_h"] = h
        store_x = N <= 10
        if store_x:
            arrays[key + "_x"] = x
        meta.append({"id": key, "N": N, "taps": list(taps), "kind": kind, "R": R, "seed": seed,
                     "shape": shape, "dtype": dtype, "x_sha256": inputs.digest(x),
                     "x_stored": store_x, "avg_dtype": avg_dtype,
                     "peak": float(np.max(np.abs(h))), "argmax": int(np.argmax(np.abs(h)))})
        print("IR small", key, N, kind, R, dtype, flush=True)
    np.savez_compressed(os.path.join(HERE, "ir_small.npz"), **arrays)
    write_json("ir_small.json", {"cases": meta})

    large = [
        (16, TAPS[16][0], "system", 2, 11, "col", "float64"),   # BASELINE config 1 shape
        (16, TAPS[16][0], "system", 5, 12, "col", "float64"),   # ex_mls.py rep=5
        (18, TAPS[18][0], "system", 2, 13, "col", "float64"),   # config 2 order
        (18, TAPS[18][-1], "noise", 3, 14, "flat", "float32"),
        (20, TAPS[20][0], "system", 2, 15, "col", "float64"),   # config 5 order
        (20, TAPS[20][0], "system", 17, 16, "col", "float64"),  # config 3 shape
    ]
    if not quick:
        large.append((24, TAPS[24][0], "system", 2, 17, "col", "float64"))  # config 4 order
    arrays = {}
    meta = []
    for cid, (N, taps, kind, R, seed, shape, dtype) in enumerate(large):
        t0 = time.time()
        seq = _MLS_base(N, tuple(taps)).get_full_sequence(1)[:, 0]
        x = inputs.make_input(kind, seq, R, seed, shape, dtype)
        h, avg_dtype = ref_impulse(N, taps, x)
        idx = inputs.sparse_indices(len(h), seed)
        key = "L%02d" % cid
        arrays[key + "_idx"] = idx.astype(np.int64)
        arrays[key + "_h"] = h[idx]
        # always keep the neighbourhood of the peak (the part a user looks at)
        am = int(np.argmax(np.abs(h)))
        lo = max(0, am - 64)
        arrays[key + "_peakwin"] = h[lo:lo + 256]
        meta.append({"id": key, "N": N, "taps": list(taps), "kind": kind, "R": R, "seed": seed,
                     "shape": shape, "dtype": dtype, "x_sha256": inputs.digest(x),
                     "avg_dtype": avg_dtype, "argmax": am, "peakwin_lo": lo,
                     "max": float(h[am]), "peak": float(np.max(np.abs(h))),
                     "sum": float(np.sum(h)), "sumsq": float(np.sum(h * h))})
        print("IR large", key, N, kind, R, dtype, "%.1fs" % (time.time() - t0), flush=True)
        del x, h, seq
    np.savez_compressed(os.path.join(HERE, "ir_large.npz"), **arrays)
    write_json("ir_large.json", {"cases": meta})


# --------------------------------------------------------------------------- error table
def exc_name(fn):
    try:
        fn()
    except BaseException as e:  # noqa: BLE001
        return type(e).__name__, str(e)
    return None, None


def dump_errors():
    N, taps = 5, TAPS[5][0]
    m = _MLS_base(N, taps)
    L = m.L
    good = m.get_full_sequence(3)
    table = {
        "ctor_N_none": exc_name(lambda: _MLS_base(None, taps)),
        "ctor_taps_none": exc_name(lambda: _MLS_base(N, None)),
        "ctor_taps_empty": exc_name(lambda: _MLS_base(N, ())),
        "ctor_taps_ndarray": exc_name(lambda: _MLS_base(N, np.array(taps))),
        "ctor_taps_list_ok": exc_name(lambda: _MLS_base(N, list(taps))),
        "impulse_len_eq_L": exc_name(lambda: m.get_impulse(good[:L])),
        "impulse_len_not_multiple": exc_name(lambda: m.get_impulse(np.zeros(2 * L + 3))),
        "impulse_two_channels": exc_name(lambda: m.get_impulse(np.zeros((2 * L, 2)))),
        "impulse_list": exc_name(lambda: m.get_impulse(list(good[:, 0]))),
        "impulse_ok": exc_name(lambda: m.get_impulse(good)),
        "mls_repeats_1": exc_name(lambda: MLS(N, taps, repeats=1)),
        "mls_non_minphase": exc_name(lambda: MLS(N, taps, repeats=2, B=(1, 2, 0), A=(1, 0, 0))),
        "random_taps_unknown_N": exc_name(lambda: get_random_taps(99)),
        "taps_keyerror": exc_name(lambda: TAPS[99]),
    }
    write_json("errors.json", {k: {"type": v[0], "msg": v[1]} for k, v in table.items()})


# --------------------------------------------------------------------------- class-level API goldens
def dump_api():
    arrays = {}
    meta = {}
    # MLS with an RBJ high-shelf emphasis, as in examples/ex_mls.py but small
    fs, N, rep = 48000, 10, 3
    f = zignal.filters.biquads.RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=fs)
    B, A = f.get_coefficients()
    mls = MLS(N=N, taps=TAPS[N][0], fs=fs, repeats=rep, B=B, A=A)
    meta["mls"] = {"N": N, "fs": fs, "repeats": rep, "taps": list(TAPS[N][0]),
                   "B": [float(b) for b in B], "A": [float(a) for a in A],
                   "samples_shape": list(mls.samples.shape), "samples_dtype": str(mls.samples.dtype),
                   "str": str(mls), "nofsamples": mls.nofsamples, "ch": mls.ch,
                   "duration": mls.duration}
    arrays["mls_raw"] = mls.samples.copy()
    mls.apply_emphasis()
    arrays["mls_emph"] = mls.samples.copy()
    # a fake system: delay by 100 samples and scale, then de-emphasis, then IR
    y = np.roll(mls.samples, 100, axis=0) * 0.5
    z = mls.apply_deemphasis(y)
    arrays["mls_deemph"] = z.copy()
    k = mls.get_impulse(z)
    assert isinstance(k, zignal.Audio)
    meta["mls"]["ir_shape"] = list(k.samples.shape)
    meta["mls"]["ir_fs"] = k.fs
    meta["mls"]["ir_ch"] = k.ch
    meta["mls"]["ir_nofsamples"] = k.nofsamples
    arrays["mls_ir"] = k.samples.copy()

    # MLS_simple loop-back, as in `python -m zignal.measure.mls` but N=10
    ms = MLS_simple(N=10, fs=96000, repeats=4)
    meta["simple"] = {"N": 10, "fs": 96000, "repeats": 4, "repr": repr(ms), "str": str(ms),
                      "samples_shape": list(ms.samples.shape)}
    arrays["simple_samples"] = ms.samples.copy()
    ir = ms.get_impulse(ms.samples)
    arrays["simple_ir"] = ir.samples.copy()

    # all nine RBJ types (MLS users pick their own emphasis filter)
    rbj = {}
    for ft in ("lowpass", "highpass", "bandpass1", "bandpass2", "notch", "allpass", "peak",
               "lowshelf", "highshelf"):
        ff = zignal.filters.biquads.RBJ(filtertype=ft, gaindb=-6.5, f0=1234.0, Q=1.3, fs=44100)
        b, a = ff.get_coefficients()
        rbj[ft] = {"B": [float(v) for v in b], "A": [float(v) for v in a],
                   "minphase": bool(ff.is_minimum_phase())}
    meta["rbj"] = {"gaindb": -6.5, "f0": 1234.0, "Q": 1.3, "fs": 44100, "types": rbj}
    np.savez_compressed(os.path.join(HERE, "api.npz"), **arrays)
    write_json("api.json", meta)


def dump_simple_short():
    """MLS_simple.get_impulse (de-emphasis + impulse response, mls.py:454-458) at orders whose period is far shorter
    than the emphasis filter's memory (100 Hz shelf at 48 kHz: thousands of samples against L = 31 ... 1023), on
    noisy, scaled captures: the period the average drops still rings into the ones it keeps."""
    arrays, cases = {}, []
    for cid, (N, rep, seed) in enumerate([(5, 4, 11), (6, 2, 12), (7, 3, 13), (8, 9, 14), (10, 6, 15)]):
        ms = MLS_simple(N=N, fs=48000, repeats=rep)
        rng = np.random.default_rng(seed)
        x = 0.7 * np.roll(ms.samples, 3, axis=0) + 0.05 * rng.standard_normal(ms.samples.shape)
        ir = ms.get_impulse(x)
        key = "s%d" % cid
        arrays[key + "_x"], arrays[key + "_ir"] = x, ir.samples.copy()
        cases.append({"id": key, "N": N, "repeats": rep, "fs": 48000, "seed": seed, "x_sha256": inputs.digest(x)})
    np.savez_compressed(os.path.join(HERE, "simple_short.npz"), **arrays)
    write_json("simple_short.json", {"cases": cases})


def dump_fft():
    """Audio.fft (audio.py:678-704) on decaying-noise 'impulse responses': an ordinary size, a Mersenne-prime size
    (fftsize bumped by one), the MLS use (rectangular, not normalised) and the default (hann, normalised)."""
    arrays, cases = {}, []
    for cid, (n, ch, fs, window, normalise, seed) in enumerate([
            (1023, 1, 48000, "rectangular", False, 1), (1023, 2, 48000, "hann", True, 2),
            (8191, 1, 96000, "rectangular", False, 3), (8191, 2, 44100, "hamming", True, 4),
            (4095, 3, 48000, "rectangular", True, 5)]):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((n, ch)) * np.exp(-np.arange(n) / (n / 8.0))[:, None]
        a = zignal.Audio(fs=fs, initialdata=x)
        frq, mag = a.fft(window=window, normalise=normalise)
        key = "c%d" % cid
        arrays[key + "_frq"], arrays[key + "_mag"] = frq, mag
        cases.append({"id": key, "n": n, "ch": ch, "fs": fs, "window": window, "normalise": normalise, "seed": seed,
                      "bins": int(mag.shape[0]), "x_sha256": inputs.digest(x)})
    np.savez_compressed(os.path.join(HERE, "fft.npz"), **arrays)
    write_json("fft.json", {"cases": cases})


def dump_multi_source():
    """The multichannel recipe of the MLS docstring (mls.py:338-343) carried out with the reference: one MLS per output
    channel (same order, different taps), their samples appended into one Audio, all sources played together into a
    small mixing system (a short FIR + delay per (recorded channel, source) pair, plus noise), and get_impulse called
    once per pair with the MLS it is paired with. Second half: the same with an emphasis filter on every source and
    apply_deemphasis before get_impulse."""
    from scipy.signal import lfilter
    from zignal.filters.biquads import RBJ
    arrays, cases = {}, []
    for cid, (N, K, C, rep, seed, emphasis) in enumerate([(10, 3, 2, 4, 21, False), (11, 4, 3, 5, 22, False), (12, 2, 3, 2, 23, False),
                                                           (10, 3, 2, 5, 24, True), (11, 2, 3, 3, 25, True)]):
        fs = 48000
        B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=fs).get_coefficients() if emphasis else ((1, 0, 0), (1, 0, 0))
        srcs = [MLS(N=N, taps=TAPS[N][k], fs=fs, repeats=rep, B=B, A=A) for k in range(K)]
        if emphasis:
            for m in srcs:
                m.apply_emphasis()
        stim = zignal.Audio(fs=fs, initialdata=srcs[0].samples)
        stim.append(*[zignal.Audio(fs=fs, initialdata=m.samples) for m in srcs[1:]])
        rng = np.random.default_rng(seed)
        L = srcs[0].L
        x = 0.01 * rng.standard_normal((rep * L, C))
        for c in range(C):
            for k in range(K):
                fir = rng.standard_normal(12) * np.exp(-np.arange(12) / 3.0)
                delay = int(rng.integers(0, L // 2))
                # the sources repeat, so a circular shift of the steady state is what a delay does to all periods but the first
                x[:, c] += np.roll(lfilter(fir, [1.0], stim.samples[:, k]), delay)
        key = "m%d" % cid
        h = np.empty((K, L, C))
        for k in range(K):
            for c in range(C):
                col = x[:, c:c + 1]
                if emphasis:
                    col = srcs[k].apply_deemphasis(col)
                h[k, :, c] = srcs[k].get_impulse(col).samples[:, 0]
        arrays[key + "_x"], arrays[key + "_h"] = x, h
        if N == 10:
            arrays[key + "_stim"] = stim.samples.copy()
        cases.append({"id": key, "N": N, "taps": [list(TAPS[N][k]) for k in range(K)], "channels": C, "repeats": rep, "fs": fs,
                      "seed": seed, "emphasis": emphasis, "B": [float(v) for v in B], "A": [float(v) for v in A],
                      "x_sha256": inputs.digest(x)})
    np.savez_compressed(os.path.join(HERE, "multi_source.npz"), **arrays)
    write_json("multi_source.json", {"cases": cases})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    steps = {"taps": dump_taps, "errors": dump_errors, "api": dump_api, "fft": dump_fft,
             "seq": lambda: dump_seq_kats(a.quick), "prefix": dump_seq_prefix, "seq_any": dump_seq_any, "simple_short": dump_simple_short, "multi": dump_multi_source, "ir": lambda: dump_ir(a.quick)}
    for name, fn in steps.items():
        if a.only and name not in a.only.split(","):
            continue
        print("==", name, flush=True)
        fn()
    print("done")
