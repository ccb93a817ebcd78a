"""CPU: the host-only part of libbbq (quantizer, heap replay, error plumbing - no HIP in those translation units) built with
AddressSanitizer + UndefinedBehaviorSanitizer and with ThreadSanitizer, driven through the C ABI by tests/csrc/host_sanitize.cpp
(threads > 1 everywhere).  GPU sanitizers are not available on this pool (sanitizers on the CPU build only)."""
import os
import shutil
import subprocess

import pytest

from bbqlib import ROOT

CSRC = os.path.join(ROOT, "better-binary-quantization_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "csrc", "host_sanitize.cpp"), os.path.join(CSRC, "bbq_quantizer.cpp"), os.path.join(CSRC, "bbq_replay.cpp")]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]), ("tsan", ["-fsanitize=thread"])])
def test_host_code_under_sanitizers(tmp_path, name, flags):
    exe = str(tmp_path / ("host_" + name))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-pthread"] + flags + SOURCES + ["-o", exe]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "host sanitize ok" in r.stdout
    assert "Sanitizer" not in r.stdout, r.stdout[-4000:]
