"""AddressSanitizer run of the host side of the C ABI: `make -C csrc asan` builds the library with the
host code instrumented (device code as usual - GPU sanitizers are not available on this pool), then tests/asan_abi_driver.py
runs under the preloaded ASAN runtime. CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.slow
def test_host_side_of_the_abi_is_asan_clean():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    rt = subprocess.check_output([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no shared ASAN runtime in this toolchain")
    lib = os.path.join(ROOT, "build", "libdensepose_hip_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "densepose_torchscript_amd", "csrc"), "asan", "ASAN_OUT=" + lib])
    env = dict(os.environ, LD_PRELOAD=rt, DP_HIP_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_abi_driver.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "asan driver ok" in out, out[-3000:]
    assert "AddressSanitizer" not in out, out[-3000:]
