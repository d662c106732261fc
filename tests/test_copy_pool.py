"""csrc/copy_pool.hpp -- the threads that fill and empty the pinned bounce buffers of gft_scan / gft_process from host
memory -- compiled on its own and run under ThreadSanitizer (plain when the toolchain cannot): the host-memory entry points
themselves need a device, the pool's hand-off between generations does not."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_copy_pool_under_tsan(tmp_path):
    src = os.path.join(ROOT, "tests", "cxx", "copy_pool_test.cpp")
    inc = os.path.join(ROOT, "gofindthem_amd", "csrc")
    exe = str(tmp_path / "copy_pool_test")
    base = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", inc, src, "-o", exe]
    tsan = subprocess.run(base + ["-fsanitize=thread"], capture_output=True, text=True)
    if tsan.returncode != 0:
        subprocess.check_call(base)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    env.pop("LD_PRELOAD", None)             # (tools/asan_host.sh preloads the ASan runtime: not into a TSan binary)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    if tsan.returncode == 0 and r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        # (ThreadSanitizer cannot map its shadow under this kernel's address-space layout: run the plain build instead)
        subprocess.check_call(base)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "copy pool ok" in r.stdout
