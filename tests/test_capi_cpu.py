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


def test_table_range_reduction_is_a_bijection():
    """The record table has ANY number of buckets (no power-of-two jumps in its size): home = top-q-bits * nbuckets >> q.  For the
    table to stay lossless, (home, remainder) must identify the 64-bit hash: checked against an independent Python restatement and
    by the inverse the export kernel uses (engine.h: table_slot / table_hash_of; pure host arithmetic, no GPU)."""
    import ctypes as C
    import numpy as np
    L = slacken_amd.lib()
    rng = np.random.default_rng(11)
    sizes = [32, 33, 63, 64, 65, 1000, 12345, 2**20, 2**20 + 1, 3 * 2**19, 2**31 - 1, 2**31, 2**31 + 12345, 2**32 - 1, 2**32,
             int(8.93e8), int(1.12e9)] + [int(v) for v in rng.integers(32, 2**32, 20)]
    for nb in sizes:
        q = max(5, (nb - 1).bit_length())
        hs = [int(v) for v in rng.integers(0, 2**64, 300, dtype=np.uint64)] + [0, 2**64 - 1, 2**63, 2**(64 - q), 2**(64 - q) - 1]
        seen = {}
        for h in hs:
            home, rem, back = C.c_uint32(), C.c_uint64(), C.c_uint64()
            assert L.slk_table_slot(nb, h, C.byref(home), C.byref(rem)) == 0
            # restatement: x = top q bits; home = x * nb >> q; extra tells the (at most) two x of one home apart
            x = h >> (64 - q)
            prod = x * nb
            want_home, extra = prod >> q, int((prod & ((1 << q) - 1)) >= nb)
            want_rem = (h & ((1 << (64 - q)) - 1)) | (extra << (64 - q))
            assert (home.value, rem.value) == (want_home, want_rem)
            assert home.value < nb and rem.value < (1 << (64 - q + 1))
            if nb == 1 << q:
                assert extra == 0 and home.value == x      # a power of two: the plain prefix
            assert L.slk_table_hash_of(nb, home.value, rem.value, C.byref(back)) == 0
            assert back.value == h
            assert seen.setdefault((home.value, rem.value), h) == h
    # every bucket of a small table is some hash's home, and consecutive top-bit values never skip one
    nb, q = 1000, 10
    homes = set()
    for x in range(1 << q):
        home, rem = C.c_uint32(), C.c_uint64()
        L.slk_table_slot(nb, x << (64 - q), C.byref(home), C.byref(rem))
        homes.add(home.value)
    assert homes == set(range(nb))
