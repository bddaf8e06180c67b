"""CPU: libfishrt.so loads and exports every entry point include/fishrt.h declares (no compute calls without a GPU);
the product package never imports the oracle; handle creation fails loudly without a device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, has_gpu

import fishrt


def _declared():
    src = open(os.path.join(ROOT, "include", "fishrt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", src))
    return names - {"fs_frame_cb"}


def test_header_symbols_exported():
    L = fishrt.lib()
    declared = _declared()
    assert declared == set(fishrt.SYMBOLS), declared ^ set(fishrt.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), f"libfishrt.so does not export {s}"
    assert b"gfx950" in L.fs_version()


def test_struct_layouts_match_header():
    from fishrt import _ffi
    assert ctypes.sizeof(_ffi.ModelArgs) == 14 * 4
    assert ctypes.sizeof(_ffi.TokenCfg) == 5 * 4
    assert ctypes.sizeof(_ffi.Sampling) == 32
    assert ctypes.sizeof(_ffi.GenStats) == 64


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fish-speech.rs_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle/" not in txt.replace("test oracle", "") and "import oracle" not in txt and "liboracle" not in txt, fn


@pytest.mark.skipif(has_gpu(), reason="CPU-only behaviour")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        fishrt.DualARTransformer(fishrt.config.TINY, fishrt.config.TINY_TOKENS)
    with pytest.raises(RuntimeError):
        fishrt.FireflyCodec()
