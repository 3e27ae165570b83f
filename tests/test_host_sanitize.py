"""ASAN + UBSAN pass over the host side of the C ABI: csrc/core.cpp is compiled unchanged with
-fsanitize=address,undefined and linked with tests/native/host_sanitize.cpp, which stands in for the HIP runtime and for the
kernels and checks, for a few thousand random clip configurations run through the whole call sequence, that every buffer a kernel
would be handed lies inside the bound workspace and that strips x segments cover each level.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_core_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", os.path.join(ROOT, "colorvideovdp_amd", "csrc", "core.cpp"),
           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), "-o", str(exe)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "sanitize" in p.stderr and "cannot find" in p.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert p.returncode == 0, p.stderr[-3000:]
    for seed in (1, 2):
        r = subprocess.run([str(exe), "1500", str(seed)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert "0 findings" in r.stdout
