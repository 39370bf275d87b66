#!/usr/bin/env python3
"""Headline benchmark: batched MLS impulse-response recovery at N=20 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 3, one GPU's share): order N=20, 512 channels x 17 periods
(first discarded, 16 averaged), float64, synthetic captures generated on the device in zignal's
(Nsamples, Nchannels) layout. A "step" is one pass of the whole hot path over that batch:
gather + average + FWHT + scatter + /L, all channels. Inputs are resident in HBM before the timed
region. With N > 1 every rank owns its own 512 channels (captures are independent: no data-path
collective, weak scaling). The ranks are launched by the driver's torch.distributed.run or by
`bench.py --gpus N` itself (N children, one per GPU); either way a rank only reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT from its environment. Barrier, max over ranks and the RCCL id bytes go over
zignal_amd.rendezvous (localhost TCP, standard library): nothing here imports PyTorch.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic bytes of one step / device time of its kernels (HIP events), vs 8 TB/s,
                 beside what hand-written stream kernels read and copy on this GPU in the same run
  cpu_baseline : the oracle (Python restatement of the reference) timed on this box's host cores
                 on a bounded sample of the same workload
  configs      : (N=1, default workload) the other BASELINE.json configurations that fit one GPU, a few
                 steps each with their own roofline, parity against the oracle and CPU baseline
  order_sweep  : (with configs) N = 16..24 on the same 10.7 GB of capture each (2^(28-N) channels x 4 averaged periods)
  next_rows    : (with configs) SURVEY §8(f) rows, device time per call: magnitude response, emphasis filter, fused de-emphasis
  period_mode  : (N>1) the same batch with the PERIODS striped over the GPUs instead of the channels:
                 partial period sums reduce-scattered with RCCL over xGMI (BASELINE config 3's
                 "RCCL reduce"); `value` stays the channel-sharded (collective-free) number.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from zignal_amd import _lib, multi  # noqa: E402
from zignal_amd.rendezvous import Rendezvous  # noqa: E402
from zignal_amd.device import DeviceBuffer, Plan, Timer, download_channel, sync  # noqa: E402
from zignal_amd.measure.mlstaps import TAPS  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the guide's measured copy rate
SEED = 0x5A17
CODE = {"f64": _lib.ZMLS_F64, "f32": _lib.ZMLS_F32}
ESIZE = {"f64": 8, "f32": 4}
TOL = {"f64": 1e-12, "f32": 1e-5}      # BASELINE.json north_star: max-abs error relative to peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--order", type=int, default=20)
    ap.add_argument("--channels", type=int, default=512, help="channels per GPU")
    ap.add_argument("--periods", type=int, default=16, help="averaged periods (capture holds one more)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="frame", choices=["frame", "capture"],
                    help="frame: (R*L, C) zignal Audio layout; capture: (C, R*L) stack of single-channel captures")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--mode", default="channel", choices=["channel", "period"],
                    help="channel: every GPU owns its own channels, no exchange (default); period: every GPU holds "
                         "some periods of ALL channels, partial sums are reduce-scattered with RCCL (SURVEY.md 8(e))")
    ap.add_argument("--chunk-channels", type=int, default=64, help="period mode: channels of a rank's share per exchange chunk")
    ap.add_argument("--reserve-channels", type=int, default=0, help="workspace for this many channels only (the batch then runs in channel chunks; tuning)")
    ap.add_argument("--plain-alloc", action="store_true", help="allocate the capture with one plain hipMalloc (no placement probing)")
    ap.add_argument("--tile", type=int, default=-1, help="force a tile configuration (tuning): 0 small, 1 big, 2 mid, 3 tall, +256 no XCD swizzle")
    ap.add_argument("--resident", default="auto", help="cache-resident schedule (tuning): auto, off, or on[:MiB[:flags]]")
    ap.add_argument("--configs", default="auto", choices=["auto", "all", "none"],
                    help="the other BASELINE configs after the headline (N=1 only): auto = when the headline workload is the default one")
    ap.add_argument("--period-leg", default="auto", choices=["auto", "none"],
                    help="N>1: also run the period-sharded RCCL leg after the channel-mode timing (auto) or not (none)")
    ap.add_argument("--period-leg-steps", type=int, default=3)
    ap.add_argument("--rendezvous-timeout", type=float, default=900.0, help="seconds a rank waits for the others at a barrier before it gives up")
    ap.add_argument("--period-leg-timeout", type=float, default=240.0, help="seconds before the RCCL leg is given up (its result is then an error entry)")
    return ap.parse_args()


class quiet_stdout:
    """RCCL reports on stdout (C++ side); stdout carries the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.keep, 1)
        os.close(self.keep)


def oracle_parity(plan, d_x, d_h, n_rows, C, rs, cs, ors, ocs, code, channels, ref=None, timed_as_shipped=0, budget_s=1e9):
    """GPU responses of `channels` against the oracle on the very captures pulled back from the device. The first
    `timed_as_shipped` of them are run as the reference ships the path (stimulus regenerated by the per-bit Python loop
    on every call, zignal/measure/mls.py:207) and timed; the rest use the stimulus made once by the C half of the oracle.
    Returns (worst relative error, seconds of the as-shipped calls, number of them)."""
    from oracle import c_oracle, mls_oracle
    N, L, taps = plan.N, plan.L, plan.taps
    worst, t_used, n_timed = 0.0, 0.0, 0
    for i, c in enumerate(channels):
        x = download_channel(plan.device, d_x.ptr, code, n_rows, rs, cs, c).astype(np.float64).reshape(-1, 1)
        h_gpu = download_channel(plan.device, d_h.ptr, plan.dtype_code, L, ors, ocs, c).astype(np.float64)
        if i < timed_as_shipped and t_used < budget_s:
            t0 = time.perf_counter()
            h_cpu = mls_oracle.get_impulse(N, taps, x)[:, 0]
            t_used += time.perf_counter() - t0
            n_timed += 1
        else:
            if ref is None:
                ref = c_oracle.sequence(N, taps)
            h_cpu = mls_oracle.get_impulse(N, taps, x, ref=ref)[:, 0]
        worst = max(worst, float(np.max(np.abs(h_gpu - h_cpu)) / np.max(np.abs(h_cpu))))
    return worst, t_used, n_timed


def cpu_baseline_and_check(args, plan, d_x, d_h, n_rows, C, rs, cs, ors, ocs, dtype_code):
    """Rank 0, N=1 only. Times the oracle as shipped by the reference (sequence regenerated by the
    per-bit Python loop on every call, then fftconvolve) on channels pulled back from the device,
    and checks the GPU result of those channels against it."""
    from oracle import mls_oracle
    N, L, taps = plan.N, plan.L, plan.taps
    t_used, n_done, worst = 0.0, 0, 0.0
    chans = [0, C // 2, C - 1] + [(7 + 37 * i) % C for i in range(C)]     # spread over the batch until the budget is used
    for c in chans:
        x = download_channel(plan.device, d_x.ptr, dtype_code, n_rows, rs, cs, c).astype(np.float64).reshape(-1, 1)
        h_gpu = download_channel(plan.device, d_h.ptr, dtype_code, L, ors, ocs, c).astype(np.float64)
        t0 = time.perf_counter()
        h_cpu = mls_oracle.get_impulse(N, taps, x)[:, 0]
        t_used += time.perf_counter() - t0
        n_done += 1
        worst = max(worst, float(np.max(np.abs(h_gpu - h_cpu)) / np.max(np.abs(h_cpu))))
        if t_used >= args.cpu_seconds:
            break
    # (ii) of SURVEY.md §8(d): the same oracle with the stimulus kept between calls (FFT work only)
    ref = mls_oracle.full_sequence(N, taps, 1)
    t_c, n_c = 0.0, 0
    for c in chans[:n_done + 2]:
        x = download_channel(plan.device, d_x.ptr, dtype_code, n_rows, rs, cs, c).astype(np.float64).reshape(-1, 1)
        t0 = time.perf_counter()
        mls_oracle.get_impulse(N, taps, x, ref=ref)
        t_c += time.perf_counter() - t0
        n_c += 1
        if t_c >= args.cpu_seconds / 3:
            break
    samples = n_done * args.periods * L
    cached = {"value": n_c * args.periods * L / t_c / 1e9, "unit": "Gsamples/s", "cores": 1,
              "sample": "%d channels, %.1f s; the oracle with the stimulus generated once (trim + average + fftconvolve only)" % (n_c, t_c)}
    return {
        "value": samples / t_used / 1e9, "unit": "Gsamples/s", "cores": 1, "kind": "port",
        "sample": "%d of the %d channels of this workload (N=%d, %d averaged periods each), %.1f s; oracle/mls_oracle.py "
                  "as the reference ships it: per-bit Python LFSR regenerated every call + scipy fftconvolve"
                  % (n_done, C, N, args.periods, t_used),
        "ir_gsamples_per_s": n_done * L / t_used / 1e9,
        "sequence_cached": cached,
    }, worst


def cpu_all_cores(N, taps, R, seconds):
    """The same oracle as P independent processes (the reference is single-threaded; captures are independent), P = this
    box's CPU share: what all host cores together manage. Returns (input samples per second, P, calls, seconds)."""
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    P = max(1, min(16, share))            # a one-GPU box's share is 16 cores, whatever os.cpu_count() says
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "throughput_worker.py"), str(N), str(R), str(seconds)]
    procs = [subprocess.Popen(cmd + [str(1000 + i), ",".join(str(t) for t in taps)], stdout=subprocess.PIPE, text=True, env=env,
                              cwd=os.path.join(ROOT, "oracle")) for i in range(P)]
    calls, longest = 0, 0.0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=seconds * 6 + 120)
            c, t = out.split()
            calls += int(c)
            longest = max(longest, float(t))
        except (subprocess.TimeoutExpired, ValueError):
            pr.kill()
            return None
    L = (1 << N) - 1
    return calls * (R - 1) * L / longest, P, calls, longest


def stream_calibration(dev, d_buf):
    """What this GPU streams in this very run, with hand-written 16-byte-per-lane kernels over the capture buffer
    (zmls_dev_stream_probe; the copy overwrites the second half, so this runs after every check): a read-only sweep and
    a copy of the first half onto the second. The step reads 16 bytes for every byte it writes, so the read rate is the
    ceiling to hold it against. Returns (read GB/s, copy GB/s counting bytes read + written, bytes swept)."""
    n = min(d_buf.nbytes, 64 << 30) // 8192 * 8192
    if n < (64 << 20):
        return None
    lib = _lib.lib()
    tm = Timer(dev)
    rates = []
    for mode in (0, 1):
        best = None
        for rep in range(4):
            tm.start()
            _lib.check(lib.zmls_dev_stream_probe(dev, d_buf.ptr, n, mode, None))
            tm.stop()
            ms = tm.elapsed_ms()
            if rep and (best is None or ms < best):
                best = ms
        rates.append(n / (best * 1e-3) / 1e9)
    return rates[0], rates[1], n


def system_check(plan, d_h, dtype_code, C, ors, ocs, seed, ch_first=0, channels=None):
    """Size-independent check: the synthetic channel c is the stimulus through a known 16-tap FIR delayed by
    d_c, so the recovered response must be that FIR at lag d_c (plus averaged-down noise)."""
    worst = 0.0
    for c in (sorted({0, C // 3, C - 1}) if channels is None else channels):
        h = download_channel(plan.device, d_h.ptr, dtype_code, plan.L, ors, ocs, c).astype(np.float64)
        taps, d = plan.synth_system(ch_first + c, seed)
        # the circular autocorrelation of an MLS is 1 at lag 0 and -1/L elsewhere: a system g comes back as g(1 + 1/L) - sum(g)/L
        want = np.full(plan.L, -float(np.sum(taps)) / plan.L)
        want[(d + np.arange(16)) % plan.L] += np.asarray(taps) * (1.0 + 1.0 / plan.L)
        worst = max(worst, float(np.max(np.abs(h - want))))
    return worst


def noise_tolerance(Ravg, L):
    """the synthetic noise (sigma 1e-3 per sample) comes back divided by sqrt(periods * L): short sequences keep more of it"""
    return max(1e-4, 8 * 1e-3 / np.sqrt(Ravg * L))


def profiled_traffic(config):
    """HBM bytes per step from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE / WRITE_SIZE in separate runs
    and the gfx950 FETCH_SIZE x2 correction), if they were taken on this very workload; PMC counters cannot be collected
    from inside a timed run, so otherwise null. The latest round's summary wins."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if not name.endswith("_summary.json"):
            continue
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = d["bench_line_of_the_traced_run"]["config"]
            if all(c.get(k) == config.get(k) for k in ("order", "channels_per_gpu", "averaged_periods", "layout", "mode")) and \
                    d["bench_line_of_the_traced_run"]["dtype"] == config["dtype"]:
                t = d.get("hbm_traffic_bytes_per_step") or sum(k["hbm_traffic_bytes_per_launch"] for k in d["kernels"])
                best = (t, "profiles/" + name)
        except (KeyError, ValueError, OSError, TypeError):
            continue
    return best


def per_kernel(pass_ms, Ravg, L, es, C):
    """the bytes each pass kernel has to move by design (pass 1 reads the capture and writes the intermediate, later
    passes read and write one batch each; a single-pass plan reads the capture and writes the responses) over its own
    HIP-event time: how close each kernel is to what this part streams"""
    out = []
    merged = len(pass_ms) == 3 and pass_ms[1] == 0        # three-pass orders on the cache-resident schedule: passes 1 and 2 run chunk by chunk
    for i, ms in enumerate(pass_ms):
        if merged and i == 1:
            continue
        by = ((Ravg + 1) if i == 0 else 2) * L * es * C
        name = "pass %d" % (i + 1)
        if merged and i == 0:
            by, name = (Ravg + 3) * L * es * C, "passes 1+2 (chunks of top-bit ranges, intermediate in the Infinity Cache)"
        out.append({"kernel": name, "ms": float(ms), "bytes_by_design": int(by),
                    "GBps": by / (ms * 1e-3) / 1e9 if ms > 0 else None})
    return out


def schedule_name(launches, passes):
    """which launch order ran (csrc/api.hip::run_passes), read off the launch counts of one step"""
    first, middle, last = launches
    if passes <= 1 or first + middle + last <= 1:
        return "one pass (whole transform in one tile)"
    if first > 1:
        if middle:
            return ("cache-resident: %d chunks of top-bit ranges, each through passes 1 and 2 back to back (intermediate kept in the "
                    "Infinity Cache), then pass 3 over the whole batch" % first)
        return "cache-resident: %d chunks of channel slabs, each through both passes back to back (intermediate kept in the Infinity Cache)" % first
    return "streaming: one launch per pass over the whole batch"


def timed_passes(plan, d_x, code, n_rows, C, rs, cs, d_h, ors, ocs, reps):
    pass_ms = np.zeros(plan.info()["passes"])
    for _ in range(reps):
        pass_ms += np.array(plan.impulse_device_timed(d_x.ptr, code, n_rows, C, rs, cs, d_h.ptr, ors, ocs))
    pass_ms /= reps
    return pass_ms[:max(1, int(np.max(np.nonzero(pass_ms)[0], initial=0)) + 1)]   # N=21/22: the tall tile runs fewer passes


# --------------------------------------------------------------------------------------------- the other BASELINE configs
EXTRA_CONFIGS = [
    # key, BASELINE.json config, order, channels, averaged periods, dtype, steps, channels checked against the oracle, of which as shipped
    ("config1", "1: MLS order N=16, 1 channel, single period (ex_mls.py-style)", 16, 1, 1, "f64", 50, 1, 1),
    ("config2", "2: MLS N=18, 64-channel batch, fp64 (LDS-resident FWHT)", 18, 64, 1, "f64", 50, 3, 3),
    ("config5_f64", "5: MLS N=20 at 1024 channels, the fp64 side of the trade-off", 20, 1024, 1, "f64", 10, 3, 2),
    ("config5_f32", "5: MLS N=20 at 1024 channels, fp32 mode", 20, 1024, 1, "f32", 10, 3, 0),
    ("config4", "4: MLS N=24 (16.7M-sample sequence), multi-pass global FWHT beyond LDS, 256 channels", 24, 256, 1, "f64", 4, 2, 1),
]


def run_extra_config(spec, dev, cpu_seconds, keep):
    key, title, N, C, Ravg, dtype, steps, n_check, n_shipped = spec
    code, es = CODE[dtype], ESIZE[dtype]
    R = Ravg + 1
    plan = Plan(N, TAPS[N][0], dtype=dtype, device=dev)
    L = plan.L
    n_rows = R * L
    plan.reserve(C, staging=(C % (16 // es) != 0))
    nbytes = n_rows * C * es
    placement = "plain hipMalloc"
    if nbytes >= (1 << 30):
        # large captures are placed by the plan's search (outside the timed region); two candidates at most for the
        # ones that fill a quarter of the memory
        d_x, ms = plan.alloc_capture(n_rows, C, code, tries=2 if nbytes > (48 << 30) else 0)
        placement = "zmls_capture_alloc (gather pass %.2f ms)" % ms
    else:
        d_x = DeviceBuffer(nbytes, dev)
    d_h = DeviceBuffer(L * C * es, dev)
    plan.synth_device(d_x.ptr, code, R, C, C, 1, seed=SEED, sigma=1e-3)

    def step():
        plan.impulse_device(d_x.ptr, code, n_rows, C, C, 1, d_h.ptr, C, 1)
    for _ in range(2):
        step()
    sync(dev)
    tm = Timer(dev)
    t0 = time.perf_counter()
    tm.start()
    for _ in range(steps):
        step()
    tm.stop()
    sync(dev)
    wall_ms = (time.perf_counter() - t0) / steps * 1e3
    ms = tm.elapsed_ms() / steps
    step()
    launches = plan.last_launches()
    pass_ms = timed_passes(plan, d_x, code, n_rows, C, C, 1, d_h, C, 1, 2)
    alg = (Ravg + 1) * L * es * C
    out = {
        "baseline_config": title,
        "workload": "MLS N=%d, %d channel%s x %d averaged period%s, %s, (R*L, C) frame-major, one GPU" % (N, C, "s" * (C > 1), Ravg, "s" * (Ravg > 1), dtype),
        "order": N, "channels": C, "averaged_periods": Ravg, "dtype": dtype, "steps": steps,
        "ms_per_step": ms, "wall_ms_per_step": wall_ms, "value": C * Ravg * L / (ms * 1e-3) / 1e9, "unit": "Gsamples/s",
        "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_step": alg,
                     "pass_ms": [float(v) for v in pass_ms], "per_kernel": per_kernel(pass_ms, Ravg, L, es, C),
                     "launches_per_step": dict(zip(("first_pass", "middle_passes", "last_pass"), launches)),
                     "schedule": schedule_name(launches, len(plan.info()["pass_bits"]))},
        "capture_placement": placement,
    }
    if key == "config1":
        # the reference's own call, numpy in and numpy out through zmls_impulse_host (upload, kernels, download); measured before
        # this configuration's checks and oracle leg (measured after them, the same 40 calls took 0.28 instead of 0.11 ms each in this
        # process, copy by copy - ZMLS_TRACE_HOST=1 - and tools/config1_host_calls.py could not reproduce that outside bench.py: cause not found)
        x = np.ascontiguousarray(d_x.download((n_rows, 1), np.float64))
        plan.impulse_host(x)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            plan.impulse_host(x)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["host_api_ms_per_call"] = float(np.median(ts))
        out["host_api_ms_per_call_min_max"] = [float(min(ts)), float(max(ts))]
    chans = sorted({0, C // 2, C - 1, C // 3})[:n_check] if C > 1 else [0]
    out["system_check_max_abs_err"] = system_check(plan, d_h, code, C, C, 1, SEED, 0, channels=chans)
    if cpu_seconds > 0:
        worst, t_shipped, n_timed = oracle_parity(plan, d_x, d_h, n_rows, C, C, 1, C, 1, code, chans, timed_as_shipped=n_shipped,
                                                  budget_s=2 * cpu_seconds)
        out["parity_max_rel_err_vs_oracle"] = worst
        out["parity_channels_checked"] = len(chans)
        if n_timed:
            out["cpu_baseline"] = {"value": n_timed * Ravg * L / t_shipped / 1e9, "unit": "Gsamples/s", "cores": 1, "kind": "port",
                                   "sample": "%d channel%s of this workload in %.1f s; oracle/mls_oracle.py as the reference ships it "
                                             "(per-bit Python LFSR regenerated every call + scipy fftconvolve)" % (n_timed, "s" * (n_timed > 1), t_shipped)}
        else:
            out["cpu_baseline"] = None
    else:
        out["parity_max_rel_err_vs_oracle"] = None
        out["cpu_baseline"] = None
    if key == "config5_f64":
        keep["f64_responses"] = {c: download_channel(dev, d_h.ptr, code, L, C, 1, c).astype(np.float64) for c in sorted({0, C // 3, (3 * C) // 4, C - 1})}
        keep["f64_ms"] = ms
    if key == "config5_f32" and "f64_responses" in keep:
        ref = keep["f64_responses"]
        out["fp32_vs_fp64_rel_err"] = max(float(np.max(np.abs(download_channel(dev, d_h.ptr, code, L, C, 1, c).astype(np.float64) - h)) /
                                                np.max(np.abs(h))) for c, h in ref.items())
        out["fp32_speedup_over_fp64"] = keep["f64_ms"] / ms
        out["cpu_baseline_note"] = "the reference has no float32 mode: see config5_f64"
    d_x.free()
    d_h.free()
    plan.close()
    return out


def order_sweep(dev):
    """north_star: 'throughput is measured on synthetic captures, MLS orders N=16-24': one entry per order, the same
    10.7 GB of capture each (2^(28-N) channels x 5 periods, 4 averaged, f64, frame-major), device time of 5 steps."""
    out = []
    for N in range(16, 25):
        C, Ravg = 1 << (28 - N), 4
        R = Ravg + 1
        plan = Plan(N, TAPS[N][0], dtype="f64", device=dev)
        L = plan.L
        plan.reserve(C)
        d_x, d_h = DeviceBuffer(R * L * C * 8, dev), DeviceBuffer(L * C * 8, dev)
        plan.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=SEED, sigma=1e-3)
        for _ in range(2):
            plan.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
        sync(dev)
        tm = Timer(dev)
        tm.start()
        for _ in range(5):
            plan.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
        tm.stop()
        ms = tm.elapsed_ms() / 5
        alg = (Ravg + 1) * L * 8 * C
        info = plan.info()
        passes = info["tall_passes"] or info["passes"]
        out.append({"order": N, "channels": C, "averaged_periods": Ravg, "passes": passes, "ms_per_step": ms,
                    "value": C * Ravg * L / (ms * 1e-3) / 1e9, "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "system_check_max_abs_err": system_check(plan, d_h, _lib.ZMLS_F64, C, C, 1, SEED, 0)})
        d_x.free()
        d_h.free()
        plan.close()
    return out


# --------------------------------------------------------------------------------------------- the RCCL leg
def next_rows(dev):
    """the rows either side of the hot path, device time per call on synthetic data (parity: tests/test_gpu_iir.py,
    test_gpu_spectrum.py, test_gpu_sources.py; tools/secondary_bench.py prints more shapes). Never part of `value`."""
    from zignal_amd.device import IirFilter, Spectrum

    def timed(fn, reps):
        fn(); sync(dev)
        t = Timer(dev)
        t.start()
        for _ in range(reps):
            fn()
        t.stop(); sync(dev)
        return t.elapsed_ms() / reps

    rows = {}
    for N, C in ((20, 1), (20, 64)):
        L = (1 << N) - 1
        sp = Spectrum(L, C, device=dev)
        h = DeviceBuffer(L * C * 8, dev); h.fill(0)
        m = DeviceBuffer(sp.n_bins * C * 8, dev)
        rows["spectrum_N%d_x%d_ms" % (N, C)] = timed(lambda: sp.magnitude_db_device(h.ptr, C, 1, None, False, m.ptr), 5)
        sp.close(); h.free(); m.free()
    N, C, R = 20, 512, 17
    L = (1 << N) - 1
    from zignal_amd.filters.biquads import RBJ
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()   # MLS_simple's emphasis filter (mls.py:433)
    f = IirFilter(A, B, device=dev)                              # de-emphasis: its inverse (mls.py:365-366)
    p = Plan(N, TAPS[N][0], device=dev)
    p.reserve(C, staging=True)
    x = DeviceBuffer(R * L * C * 8, dev)
    p.synth_device(x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=SEED + 77, sigma=1e-3)      # the headline's kind of capture (a zero-filled one streams ~1 ms faster)
    y = DeviceBuffer(2 * L * C * 8, dev)
    f.reserve(2 * L, C)
    rows["filter_2L_x512_ms"] = timed(lambda: f.filter_device(x.ptr, _lib.ZMLS_F64, 2 * L, C, C, 1, y.ptr, C, 1), 3)
    y.free()
    hh = DeviceBuffer(L * C * 8, dev)
    f.reserve_fused(p, R * L, C)
    rows["deemphasis_fused_with_impulse_N20_x512_x17_ms"] = timed(lambda: p.impulse_filtered_device(f, x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, hh.ptr, C, 1), 3)
    rows["deemphasis_fused_form"] = ("filter after the transform" if f.last_fused_form() == 1 else "one sweep in time order") + " (DESIGN.md §9)"
    rows["impulse_same_capture_ms"] = timed(lambda: p.impulse_device(x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, hh.ptr, C, 1), 3)
    x.free(); hh.free(); f.close(); p.close()
    rows["what"] = ("device time per call, synthetic data: magnitude response (Audio.fft, own transform passes) of N=20 responses; the emphasis "
                    "filter over 2 periods x 512 channels; MLS_simple.get_impulse's de-emphasis + impulse response in one call on 17 periods x "
                    "512 channels, and the plain impulse response of the same (plainly allocated) capture beside it")
    return rows


def period_leg(args, world, rank, dev, rdzv, plan, dtype_code, es, C, R, result):
    """The same batch with the periods, not the channels, spread over the ranks (second mode): rank r
    holds periods r, r+world, ... of ALL world*C channels, adds them up, the partial sums are reduce-scattered with RCCL
    (zmls_impulse_period_sharded_device), and rank r transforms channel block r. Fills `result` in place."""
    L = plan.L
    Ct = world * C
    mine, skip, total = multi.period_shard(R, world, rank)
    n_rows = len(mine) * L
    chunk = min(args.chunk_channels, C)
    state = {}

    def agreed(what, fn):
        """run fn on every rank; all ranks learn whether it worked everywhere (a rank that went on alone would hang the rest)"""
        ok, err = 1.0, ""
        try:
            fn()
        except Exception as e:  # noqa: BLE001  (RCCL refuses e.g. two ranks on one device)
            ok, err = 0.0, "%s: %s" % (type(e).__name__, e)
        if rdzv.min(ok) == 0.0:
            result["error"] = "%s failed over %d ranks: %s" % (what, world, err or "on another rank")
            for k in ("d_x", "d_h"):
                if k in state:
                    state[k].free()
            if "comm" in state:
                state["comm"].close()
            return False
        return True

    def which_rccl():
        state["lib"] = multi.rccl_library()

    def make_comm():
        blob = rdzv.broadcast(multi.make_unique_id)
        with quiet_stdout():
            state["comm"] = multi.Comm(blob, rank, world, dev)

    def make_buffers():
        state["d_x"] = DeviceBuffer(max(n_rows, 1) * Ct * es, dev)
        state["d_h"] = DeviceBuffer(L * C * es, dev)
        state["ps"] = multi.PeriodSharded(plan, state["comm"], chunk)
        if mine:
            plan.synth_device(state["d_x"].ptr, dtype_code, len(mine), Ct, Ct, 1, seed=SEED, sigma=1e-3)
        sync(dev)

    if not agreed("loading librccl", which_rccl):
        return
    # which RCCL answered, on every rank (a first multi-GPU record should explain itself)
    result["librccl_path"], result["librccl_version"] = state["lib"]
    if not agreed("ncclCommInitRank", make_comm) or not agreed("staging for the exchange", make_buffers):
        return
    comm, d_x, d_h, ps = state["comm"], state["d_x"], state["d_h"], state["ps"]
    result["rccl_ranks"] = comm.info()["world"]

    def step():
        ps.impulse_device(d_x.ptr, dtype_code, n_rows, skip, total, Ct, Ct, 1, d_h.ptr, C, 1)
    step()
    sync(dev)
    rdzv.barrier()
    steps = max(1, args.period_leg_steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(dev)
    mine_ms = (time.perf_counter() - t0) / steps * 1e3         # this rank alone, before it waits for the others
    rdzv.barrier()
    per_rank = rdzv.gather_floats(mine_ms)
    ms = rdzv.max((time.perf_counter() - t0) / steps * 1e3)
    err_sys = rdzv.max(system_check(plan, d_h, dtype_code, C, C, 1, SEED, rank * C))
    sent = (world - 1) * C * L * es                     # bytes of partial sums that leave each GPU per step
    result.update({
        "ms_per_step": ms, "steps": steps, "chunk_channels": comm.info()["chunk_channels"],
        "ms_per_step_per_rank": {"min": min(per_rank), "max": max(per_rank), "all": per_rank},
        "value": Ct * total * L / (ms * 1e-3) / 1e9, "unit": "Gsamples/s",
        "bytes_exchanged_per_gpu_per_step": sent, "GBps_exchanged_per_gpu": sent / (ms * 1e-3) / 1e9,
        "periods_held_by_rank0": len(multi.period_shard(R, world, 0)[0]),
        "system_check_max_abs_err": err_sys,
        "workload": "the same %d channels x %d averaged periods, periods striped over %d GPUs; every rank adds up its periods, "
                    "ncclReduceScatter(sum) of the partial sums in chunks of %d channels, rank r transforms channel block r"
                    % (Ct, total, world, comm.info()["chunk_channels"]),
    })
    d_x.free()
    d_h.free()
    comm.close()


def launch_ranks(n):
    """`bench.py --gpus N` started by hand: N children of this process, one per GPU, before anything here has touched a
    GPU (a process that has must never be replaced, and is not: the children are ordinary subprocesses). Rank 0's stdout
    is ours; the exit status is the worst of theirs."""
    port = os.environ.get("MASTER_PORT", "29517")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    for pr in procs:
        rc = pr.wait()
        worst = worst or rc
    return worst


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    # barrier, max over ranks, id bytes: localhost TCP among the ranks of this launch (no PyTorch, no gloo)
    rdzv = Rendezvous(rank, world, timeout_s=args.rendezvous_timeout)
    rdzv.barrier()

    N, C, Ravg = args.order, args.channels, args.periods
    R = Ravg + 1
    n_dev = _lib.device_count()
    if n_dev < 1:
        print("no GPU visible: libzmls has no CPU path", file=sys.stderr)
        sys.exit(1)
    dev = local_rank % n_dev          # fewer GPUs than ranks only happens when rehearsing the launch on a small box
    dtype_code, es = CODE[args.dtype], ESIZE[args.dtype]
    t0 = time.perf_counter()
    plan = Plan(N, TAPS[N][0], dtype=args.dtype, device=dev)
    plan_build_s = time.perf_counter() - t0
    L = plan.L
    n_rows = R * L
    if args.layout == "frame":
        rs, cs, ors, ocs = C, 1, C, 1
    else:
        rs, cs, ors, ocs = 1, n_rows, 1, L
    d_h = DeviceBuffer(L * C * es, dev)
    placement = None
    if args.tile >= 0:
        plan.force_tile(args.tile)
    if args.resident != "auto":
        r = args.resident.split(":")
        plan.set_resident({"off": 0, "on": 1}[r[0]], (int(r[1]) << 20) if len(r) > 1 else 0, int(r[2]) if len(r) > 2 else 0)
    comm = None
    if args.mode == "channel":
        seed, ch_first = SEED + 1000 * rank, 0
        plan.reserve(args.reserve_channels or C, staging=(args.layout == "capture"))
        d_x = None
        if args.layout == "frame" and not args.plain_alloc:
            # placement of the capture in physical memory is chosen by probing (zmls_capture_alloc): outside the timed region
            try:
                d_x, st = plan.alloc_capture(n_rows, C, dtype_code, stats=True)
                if st["candidates_tried"]:
                    placement = st
            except _lib.ZmlsError as e:      # a plain buffer still measures
                print("capture placement search failed (%s); plain allocation" % e, file=sys.stderr)
        if d_x is None:
            d_x = DeviceBuffer(n_rows * C * es, dev)
        plan.synth_device(d_x.ptr, dtype_code, R, C, rs, cs, seed=seed, sigma=1e-3)

        def step():
            plan.impulse_device(d_x.ptr, dtype_code, n_rows, C, rs, cs, d_h.ptr, ors, ocs)
    else:
        # every rank holds periods rank, rank+world, ... of ALL world*C channels; it ends up with the responses
        # of channel block `rank`. Per-GPU bytes are those of the channel mode (weak scaling).
        assert args.layout == "frame", "period mode is benchmarked in the frame-major layout"
        seed, ch_first = SEED, rank * C
        Ct = world * C
        mine, skip, total = multi.period_shard(R, world, rank)
        n_rows = len(mine) * L
        d_x = DeviceBuffer(max(n_rows, 1) * Ct * es, dev)
        blob = rdzv.broadcast(multi.make_unique_id)
        with quiet_stdout():
            comm = multi.Comm(blob, rank, world, dev)
        ps = multi.PeriodSharded(plan, comm, min(args.chunk_channels, C))
        if mine:
            plan.synth_device(d_x.ptr, dtype_code, len(mine), Ct, Ct, 1, seed=seed, sigma=1e-3)

        def step():
            ps.impulse_device(d_x.ptr, dtype_code, n_rows, skip, total, Ct, Ct, 1, d_h.ptr, C, 1)
    sync(dev)

    def barrier():
        sync(dev)
        rdzv.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    tm = Timer(dev)                    # HIP events on the launch stream, around the very steps the wall clock times
    t0 = time.perf_counter()
    tm.start()
    for _ in range(args.steps):
        step()
    tm.stop()
    sync(dev)
    elapsed = time.perf_counter() - t0
    pipeline_ms = tm.elapsed_ms() / args.steps
    per_rank_ms = rdzv.gather_floats(elapsed / args.steps * 1e3)     # every rank's own time; the job's is the slowest
    elapsed = max(per_rank_ms) * args.steps / 1e3
    per_rank_kernel_ms = rdzv.gather_floats(pipeline_ms)

    # ---- kernel-level timing on the launch stream (HIP events around every pass kernel)
    pass_ms = np.zeros(plan.info()["passes"])
    if args.mode == "channel":
        pass_ms = timed_passes(plan, d_x, dtype_code, n_rows, C, rs, cs, d_h, ors, ocs, max(3, min(args.steps, 10)))

    alg_bytes = (Ravg + 1) * L * es * C                    # (R_avg+1)*L*sizeof(T) per channel
    achieved = alg_bytes / (pipeline_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "MLS IR deconvolution Gsamples/s + %% HBM roofline, N=%d, 1/2/4/8 MI355X" % N,
        "value": world * C * Ravg * L * args.steps / elapsed / 1e9,
        "unit": "Gsamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "MLS N=%d, %d channels x %d averaged periods per GPU (BASELINE config 3: 4096 captures x 16 "
                        "periods over 8 GPUs), %s, layout %s" % (N, C, Ravg, args.dtype,
                        "(R*L, C) frame-major (zignal Audio)" if args.layout == "frame" else "(C, R*L) capture-major"),
            "order": N, "channels_per_gpu": C, "averaged_periods": Ravg, "layout": args.layout,
            "parallelism": ("channels sharded over %d GPU(s), no data-path collective" % world) if args.mode == "channel" else
                           ("periods striped over %d GPU(s), RCCL reduce-scatter of partial period sums in chunks of %d channels"
                            % (world, min(args.chunk_channels, C))),
            "mode": args.mode,
            "taps": list(TAPS[N][0]),
        },
        "ir_gsamples_per_s": world * C * L * args.steps / elapsed / 1e9,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": "fwht_pass_kernel x%d (one launch sequence = one step)" % len(pass_ms),
            "alg_bytes_per_step": alg_bytes, "kernel_ms_per_step": pipeline_ms,
            "pass_ms": [float(v) for v in pass_ms],
            "per_kernel": per_kernel(pass_ms, Ravg, L, es, C),
            # the cache-resident schedule runs one step as several chunk launches of each pass kernel (csrc/api.hip::run_passes)
            "launches_per_step": dict(zip(("first_pass", "middle_passes", "last_pass"), plan.last_launches())),
            "schedule": schedule_name(plan.last_launches(), len(pass_ms)),
        },
        "plan_build_s": plan_build_s,
        # the spread over the ranks (placement plateaus differ from GPU to GPU: DESIGN.md §3): wall time per step of every
        # rank, and the HIP-event time of its launch sequences
        "per_rank": {"ms_per_step": per_rank_ms, "kernel_ms_per_step": per_rank_kernel_ms,
                     "min_ms": min(per_rank_ms), "max_ms": max(per_rank_ms)},
        # what the placement search of the capture bought: gather-pass ms of the first candidate (= one plain hipMalloc
        # in this process) against the buffer the steps above ran on
        "capture_placement": placement,
    }
    out["system_check_max_abs_err"] = system_check(plan, d_h, dtype_code, C, ors, ocs, seed, ch_first)
    if not args.no_check and out["system_check_max_abs_err"] > noise_tolerance(Ravg, L):
        print("SYSTEM CHECK FAILURE on rank %d: %.3e" % (rank, out["system_check_max_abs_err"]), file=sys.stderr)
        sys.exit(3)
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and args.mode == "channel":
        cpu, worst = cpu_baseline_and_check(args, plan, d_x, d_h, n_rows, C, rs, cs, ors, ocs, dtype_code)
        out["cpu_baseline"] = cpu
        out["parity_max_rel_err_vs_oracle"] = worst
        par = cpu_all_cores(N, plan.taps, R, min(8.0, args.cpu_seconds))
        if par:
            cpu["all_cores"] = {"value": par[0] / 1e9, "unit": "Gsamples/s", "cores": par[1],
                                "sample": "%d oracle calls on capture-shaped noise (N=%d, %d averaged periods) by %d independent "
                                          "processes in %.1f s" % (par[2], N, Ravg, par[1], par[3])}
        if not args.no_check and worst > TOL[args.dtype]:
            print("PARITY FAILURE: %.3e > %.1e" % (worst, TOL[args.dtype]), file=sys.stderr)
            sys.exit(2)
    else:
        out["cpu_baseline"] = None
    cal = stream_calibration(dev, d_x)
    if cal:
        roof = out["roofline"]
        roof["stream_read_GBps"], roof["stream_copy_GBps"] = cal[0], cal[1]
        roof["stream_calibration"] = ("hand-written 16 B/lane kernels over %.1f GB of the capture buffer in this run: read-only sweep, "
                                      "and copy of one half onto the other (bytes read + written)" % (cal[2] / 1e9))
        roof["frac_of_same_run_read_rate"] = achieved / cal[0]
        # what the step would take if every stream ran at the rate of the calibration kernel of its kind: the capture read at the
        # read-only rate; the intermediate (written, read back) and the responses (written) at the copy rate, which counts bytes
        # read + written. Two passes only (N <= 20): the intermediate makes one round trip.
        if len(pass_ms) == 2 and args.mode == "channel":
            cells = L * C * es
            bound_ms = (Ravg * cells / cal[0] + 3 * cells / cal[1]) / 1e6
            roof["same_run_streaming_bound_ms"] = bound_ms
            roof["frac_of_same_run_streaming_bound"] = bound_ms / pipeline_ms
    prof = profiled_traffic(dict(out["config"], dtype=args.dtype))
    if prof:
        out["roofline"]["traffic"] = prof[0]
        out["roofline"]["traffic_source"] = "%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per step, all pass kernels)" % prof[1]

    # ---- everything below runs on memory the headline has given back
    d_x.free()
    d_h.free()
    if comm is not None:
        comm.close()

    if world > 1 and args.mode == "channel" and args.period_leg == "auto":
        # the RCCL leg must never cost the line above: it runs beside a deadline, and a rank that finds it stuck
        # reports that and leaves
        result = {}
        th = threading.Thread(target=lambda: _guard(period_leg, result, args, world, rank, dev, rdzv, plan, dtype_code, es, C, R, result), daemon=True)
        th.start()
        th.join(args.period_leg_timeout)
        if th.is_alive():
            result = {"error": "the period-sharded leg did not finish within %.0f s" % args.period_leg_timeout}
            out["period_mode"] = result
            if rank == 0:
                print(json.dumps(out), flush=True)
            # the line is out; the status still says that a leg hung (a daemon thread may be stuck inside RCCL: no clean-up)
            sys.stderr.write("rank %d: %s\n" % (rank, result["error"]))
            sys.stderr.flush()
            os._exit(4)
        out["period_mode"] = result
    plan.close()

    run_configs = args.configs == "all" or (args.configs == "auto" and (N, C, Ravg, args.dtype, args.layout, args.mode) == (20, 512, 16, "f64", "frame", "channel"))
    if rank == 0 and world == 1 and run_configs:
        keep, extra = {}, []
        for spec in EXTRA_CONFIGS:
            try:
                extra.append(run_extra_config(spec, dev, args.cpu_seconds, keep))
            except Exception as e:  # noqa: BLE001  (the headline must survive a failure here; the entry says what happened)
                extra.append({"baseline_config": spec[1], "error": "%s: %s" % (type(e).__name__, e)})
        out["configs"] = extra
        try:
            out["order_sweep"] = order_sweep(dev)
        except Exception as e:  # noqa: BLE001
            out["order_sweep"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["next_rows"] = next_rows(dev)
        except Exception as e:  # noqa: BLE001
            out["next_rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
        bad = [c.get("workload", c["baseline_config"]) for c in extra if "error" in c
               or (c.get("parity_max_rel_err_vs_oracle") or 0) > TOL[c["dtype"]]
               or c["system_check_max_abs_err"] > noise_tolerance(c["averaged_periods"], (1 << c["order"]) - 1)]
        # a failure here is reported in the line and on stderr; it does not take the headline (checked above) with it
        out["configs_failed"] = bad
        if bad:
            print("CONFIG FAILURE: %s" % json.dumps(bad)[:2000], file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    try:
        rdzv.barrier()
    except (TimeoutError, ConnectionError, OSError) as e:      # a rank that left early must not turn a printed line into a failure
        print("rank %d: %s at the closing barrier" % (rank, type(e).__name__), file=sys.stderr)
    rdzv.close()


def _guard(fn, result, *a):
    try:
        fn(*a)
    except Exception as e:  # noqa: BLE001
        result["error"] = "%s: %s" % (type(e).__name__, e)


if __name__ == "__main__":
    main()
