"""The C-ABI library loads and exports every symbol include/ipde_hip.h declares
(CPU: no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ipde_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ipde_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from ipde_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from ipde_amd.build import build_lib
        build_lib(verbose=False)
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libipde_hip.so does not export %s" % s
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms


def test_version_string():
    from ipde_amd import _lib
    assert b"gfx950" in _lib.load().ipde_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ipde_amd import _lib
    from ipde_amd.device import get_context
    with pytest.raises(_lib.IpdeHipError):
        get_context()


def test_product_does_not_import_oracle():
    """ipde_amd/ must never reference the oracle package."""
    pkg = os.path.join(ROOT, "ipde_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
