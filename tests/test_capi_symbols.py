"""CPU suite: the C-ABI library loads and exports every symbol include/sfmx.h declares (no compute calls)."""
import ctypes
import os
import re

import helpers as H


def test_libsfmx_exports_header_symbols():
    capi = H.pkg().capi if hasattr(H.pkg(), "capi") else __import__("importlib").import_module(H.PKG_NAME + ".capi")
    assert os.path.exists(capi.LIB_PATH), "libsfmx.so not built: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(capi.LIB_PATH)
    hdr = open(os.path.join(H.ROOT, "include", "sfmx.h")).read()
    declared = set(re.findall(r"^(?:int|void|void\*|double|uint64_t|const char\*)\s+(sfmx_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared, "no declarations parsed"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(lib, s), f"libsfmx.so does not export {s}"


def test_no_device_is_a_loud_error():
    """Without a GPU the product must fail, never fall back (this container has no GPU)."""
    import importlib
    import pytest
    capi = importlib.import_module(H.PKG_NAME + ".capi")
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(capi.SfmxError):
        capi.Context(0)
