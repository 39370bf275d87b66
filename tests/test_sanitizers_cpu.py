"""Sanitizer tier (CPU, no GPU call): the library's host logic and the C half of the oracle under AddressSanitizer +
UndefinedBehaviorSanitizer. GPU AddressSanitizer is not available on this pool, so device code is covered by the
geometry self checks here (every index a kernel forms is enumerated on the host) and by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_oracle_c_under_asan_and_ubsan(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "oracle", "mls_oracle.c"), os.path.join(ROOT, "oracle", "selftest.c"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "oracle selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_library_host_logic_under_asan_and_ubsan(tmp_path):
    """zmls_host.cpp (error text, GF(2) jump tables, pass planner, tile geometry check, placement search, and the transform passes of fft.hip run on the host) with the
    argument checks of api.hip in front of it; sanitizers on the host side only (-Xarch_host)."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "host_selftest")
    src = os.path.join(ROOT, "zignal_amd", "csrc")
    files = [os.path.join(src, f) for f in ("zmls_host.cpp", "api.hip", "lfsr.hip", "fwht_f64.hip", "fwht_f32.hip", "period_sum.hip",
                                             "layout.hip", "static_ws.hip", "hostcopy.hip", "fft.hip")]
    flags = ["-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
             "-Xarch_host", "-fno-sanitize-recover=all", "-DZMLS_FFT_NO_DEVICE", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
    files.append(os.path.join(ROOT, "tests", "native", "host_selftest.cpp"))
    objs = [str(tmp_path / (os.path.basename(f) + ".o")) for f in files]
    # one compiler per file, a few at a time (one invocation over all files compiles them one after the other: four minutes)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as pool:
        list(pool.map(lambda fo: subprocess.check_call([HIPCC] + flags + ["-x", "hip", "-c", fo[0], "-o", fo[1]], timeout=1500), zip(files, objs)))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined"] + objs + ["-o", exe], timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "host selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
