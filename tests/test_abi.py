"""The C-ABI shared library loads and exports exactly what include/tf_fusion.h declares.
No compute calls here (no GPU in the CPU suite)."""
import ctypes
import os
import re

import pytest

from texturefusion_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tf_fusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"TF_API\s+[\w\s\*]+?\b(tf_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in _declared() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_torch_or_oracle_in_the_product_library():
    """The boundary is a plain C ABI: the .so must not depend on torch, and must not link the oracle."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "tf_oracle" not in out
    assert "libamdhip64" in out


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "texturefusion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.findall(r"import\s+oracle|from\s+oracle|tf_oracle|tfo_\w+|oracle/|oracle\.api", src)
                assert not bad, (os.path.join(dp, f), bad)


def test_create_fails_loudly_without_a_gpu():
    L = capi.lib()
    if L.tf_device_count() > 0:
        pytest.skip("a GPU is visible: the no-device error path cannot be exercised")
    with pytest.raises(capi.TFError) as e:
        capi.Volume(0.005)
    assert e.value.code == capi.TF_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_error_codes_match_header():
    text = open(os.path.join(ROOT, "include", "tf_fusion.h")).read()
    for name in ("TF_ERR_ATLAS_FULL", "TF_ERR_INVALID", "TF_ERR_CAPACITY", "TF_ERR_HIP",
                 "TF_ERR_NO_DEVICE", "TF_ERR_MISSING_CHUNK"):
        m = re.search(r"#define\s+%s\s+\((-?\d+)\)" % name, text)
        assert m and int(m.group(1)) == getattr(capi, name)
    m = re.search(r"#define\s+TF_BOUNDARY_RECORD_BYTES\s+\(([^)]+)\)", text)
    assert eval(m.group(1)) == capi.TF_BOUNDARY_RECORD_BYTES
