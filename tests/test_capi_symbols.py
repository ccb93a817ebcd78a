"""CPU: the C-ABI library loads and exports every symbol include/bbq.h declares; device entry points fail loudly
without a GPU (no CPU fallback); the N-API addon loads under node."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from bbqlib import ROOT, bbq_amd as B, capi


def _declared():
    src = open(os.path.join(ROOT, "include", "bbq.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bbq_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libbbq.so does not export %s" % n
    assert sorted(capi.SYMBOLS) == names, "capi.py binds a different set than include/bbq.h declares"
    assert lib.bbq_abi_version() == 3


def test_no_cpu_fallback_without_device():
    if B.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(B.BBQError) as e:
        B.Index(np.zeros((4, 1), np.uint8), np.zeros((4, 4)), 8, 0.0)
    assert e.value.code == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_create_arguments_are_checked_before_the_device():
    """argument errors that do not need a device: the dimension bound of the 31-bit integer dot product (dim * 255 for packed 1-bit
    rows, dim * 255 * 255 for multi-bit fields) on both sides of the limit, and the creation options"""
    if B.device_count() > 0:
        pytest.skip("a HIP device is present: the accepted side would allocate")
    z8, z64 = np.zeros((1, 8), np.uint8), np.zeros((1, 4))

    def code_of(dim, ib, **kw):
        with pytest.raises(B.BBQError) as e:
            capi.Index(z8, z64, dim, 0.0, index_bits=ib, **kw)
        return e.value.code

    lim1 = 0x7fffffff // 255            # 8 421 504: largest 1-bit dimension
    limm = 0x7fffffff // (255 * 255)    # 33 025: largest multi-bit dimension
    assert code_of(limm + 1, 1) == capi.ERR_NO_DEVICE        # accepted for 1-bit rows (round 2 refused it): only the device is missing
    assert code_of(lim1, 1) == capi.ERR_NO_DEVICE
    assert code_of(lim1 + 1, 1) == capi.ERR_UNSUPPORTED
    assert code_of(limm, 2) == capi.ERR_NO_DEVICE
    assert code_of(limm + 1, 2) == capi.ERR_UNSUPPORTED
    assert code_of(limm + 1, 8) == capi.ERR_UNSUPPORTED
    assert code_of(64, 1, corrections=7) == capi.ERR_INVALID_ARG
    assert code_of(64, 1, corrections="inline") == capi.ERR_NO_DEVICE
    with pytest.raises(B.BBQError) as e:
        capi.Index.build(np.zeros((2, 64), np.float32), 1, corrections=-5)
    assert e.value.code == capi.ERR_INVALID_ARG


def test_product_never_links_the_oracle():
    out = subprocess.run(["ldd", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "better-binary-quantization_amd")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".c", ".py", ".js")):
                txt = open(os.path.join(root, f), encoding="utf-8", errors="replace").read()
                assert "bbq_oracle" not in txt and "orclib" not in txt, "%s references the oracle" % f


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_napi_addon_loads_and_js_host_cpu_checks():
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "cpu_checks.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 failures" in r.stdout
