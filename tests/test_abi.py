"""The C-ABI library builds, loads and exports every symbol include/meld_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "meld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(meld_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from meld_amd import _lib, build

    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libmeld_hip.so does not export {}".format(s)
    assert sorted(_lib.SIGNATURES) == syms, set(_lib.SIGNATURES) ^ set(syms)


def test_geometry_queries_and_argument_checks():
    from meld_amd import _lib

    lib = _lib.get_lib()
    assert lib.meld_abi_version() == 1
    assert lib.meld_knn_padded_dim(50) == 52
    assert lib.meld_knn_padded_dim(100) == 104
    assert lib.meld_knn_padded_dim(2) == 8
    assert lib.meld_knn_padded_dim(127) < 0
    assert b"d=127" in lib.meld_last_error()
    assert lib.meld_knn_row_capacity(64) == 128
    assert lib.meld_knn_row_capacity(129) < 0
    assert lib.meld_knn_tile_refs() == 64 and lib.meld_knn_block_queries() == 128
    # null pointers are rejected before any launch
    assert lib.meld_scale_f64(None, 1.0, None, 10, None) == -1
    assert lib.meld_cheby_step(None, None, None, None, 0, 0, 2, None, 0, None, None, None, 1.0, 0.0, 0.0, 0.0, None, None) == -1


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under meld_amd/ may import or execute it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "meld_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|meld_oracle|/root/reference", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_split_layout_geometry_is_a_function_of_the_dimension():
    """No compute, no GPU: which dimensions get the split operand layout (13 coordinates and their norm pieces in K block 0, see
    include/meld_hip.h) -- wherever d > 13 and d + 6 K slots fit the K blocks d + 3 needs; a function of d alone."""
    from meld_amd._lib import get_lib

    lib = get_lib()
    want = {1: 0, 13: 0, 14: 13, 26: 13, 27: 0, 29: 0, 30: 13, 42: 13, 43: 0, 45: 0, 46: 13, 50: 13, 58: 13, 59: 0, 61: 0, 62: 13, 141: 0}
    for d, lead in want.items():
        kb = lib.meld_knn16_kblocks(d)
        assert kb == (d + 3 + 15) // 16
        assert lib.meld_knn16_split_dims(d) == lead, d
        assert (lead == 13) == (d > 13 and d + 6 <= 16 * kb)
    assert lib.meld_knn16_split_dims(500) == 0  # (no kernel for that many K blocks)
    assert lib.meld_knn16_split_dims(50) == 13
    assert not hasattr(lib, "meld_knn16_debug_split")  # (round 5's process-wide layout switch is gone from the boundary)
    assert lib.meld_frame_max_dims() == 64
