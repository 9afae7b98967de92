"""Host-side AddressSanitizer run of the C ABI (`python -m bayesian_cbf_amd.build --asan`): the library's
launchers, argument checks and error plumbing instrumented, driven by tests/cabi_asan_driver.c on the CPU (no GPU
needed: nothing valid is launched).  GPU ASan is not available on this pool; device code is checked by the parity tests."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("BCBF_SKIP_ASAN") == "1", reason="BCBF_SKIP_ASAN=1")
def test_c_abi_under_host_address_sanitizer(tmp_path):
    from bayesian_cbf_amd import build
    lib = build.build(asan=True)                      # ~1 min the first time (all sources at -O1 -g), incremental after
    assert lib.endswith("libbcbf_asan.so") and os.path.exists(lib)
    exe = str(tmp_path / "cabi_asan_driver")
    clang = os.path.join(os.path.dirname(os.path.realpath(build._hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    cmd = [clang, os.path.join(ROOT, "tests", "cabi_asan_driver.c"), "-I" + os.path.join(ROOT, "include"),
           "-g", "-O1", "-fsanitize=address", "-shared-libsan", "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    rtdir = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               LD_LIBRARY_PATH=os.pathsep.join(filter(None, [os.path.dirname(rtdir), "/opt/rocm/lib", os.environ.get("LD_LIBRARY_PATH", "")])))
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert "AddressSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0 and "asan driver: ok" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
