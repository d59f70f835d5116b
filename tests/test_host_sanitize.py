"""CPU suite, part 4: the library's host-only code under AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer.

libmisslap's host side -- the Hopcroft-Karp matcher, the solve loop (`misslap_drive_sharded`), option / ABI-version
handling, the communicator objects, the stream / block caches -- runs without a GPU, so it can be instrumented on the
CPU box ("-fsanitize=address for host code"; GPU-side sanitizers are not available on the pool).
tests/host_sanitize.cpp is the driver; both it and misslap.hip are compiled with -fsanitize=address,undefined (device
code is left alone: -fno-gpu-sanitize).  Never run on a GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_host_entry_points_are_clean_under_asan_and_ubsan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not (os.path.exists(hipcc) and os.path.exists(CLANG)):
        pytest.skip("ROCm toolchain not found")
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("sanitizer runs belong on the CPU box")
    from sslap_amd import build
    lib = tmp_path / "libmisslap_asan.so"
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    flags = [f for f in build.FLAGS if f != "-O3"]
    subprocess.check_call([hipcc] + flags + san + ["-fno-gpu-sanitize", os.path.join(build.CSRC, "misslap.hip"), "-o", str(lib)],
                          cwd=build.CSRC)
    exe = tmp_path / "host_sanitize"
    subprocess.check_call([CLANG, "-std=c++17"] + san + ["-I", os.path.join(ROOT, "include"),
                                                         os.path.join(ROOT, "tests", "host_sanitize.cpp"), "-L", str(tmp_path),
                                                         "-lmisslap_asan", f"-Wl,-rpath,{tmp_path}", "-lpthread", "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host_sanitize ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr
