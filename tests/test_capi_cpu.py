"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/slacken_amd.h declares,
and refuses to compute without a GPU (no silent fallback)."""
import os
import re

import pytest

import slacken_amd
from slacken_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "slacken_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slk_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = slacken_amd.lib()
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/slacken_amd.h but not exported"
    assert sorted(capi.EXPORTS) == syms


def test_version_and_error_text():
    L = slacken_amd.lib()
    assert b"gfx950" in L.slk_version()
    assert isinstance(L.slk_last_error(), bytes)


def test_code_object_is_gfx950_only():
    blob = open(slacken_amd.lib_path(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in blob


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(slacken_amd.SlackenError) as e:
        slacken_amd.Index()
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slacken_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"


def test_struct_layouts():
    assert capi.SPAN_DTYPE.itemsize == 16 and capi.HIT_DTYPE.itemsize == 8


def test_header_is_plain_c(tmp_path):
    """The ABI header must be consumable from C (a JNI shim, cgo, ...): C99, no warnings."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "slacken_amd.h"\nint main(void) { slk_params p = {35, 31, 7, 1, SLK_DEFAULT_TOGGLE_MASK, 1, 0}; '
                   'slk_hit h = {0, 0}; (void)p; (void)h; return slk_device_count() < 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)])


def test_release_library_has_no_result_changing_switches():
    """The timing experiments of the hot kernel (probes off, map updates off ...) exist only in a -DSLK_TUNING build: the shipped
    library does not even hold the name of the environment variable that selects them."""
    import slacken_amd
    blob = open(slacken_amd.lib_path(), "rb").read()
    assert b"SLK_DEBUG_ABLATE" not in blob
