"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/spvipes_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "spvipes_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from spvipes_amd import _abi
    from spvipes_amd.build import build

    build()
    lib = ctypes.CDLL(_abi.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in spvipes_hip.h but not exported"
    assert set(names) == set(_abi.EXPORTED_SYMBOLS), "ctypes binding and header disagree"
    assert _abi.load().spv_version() >= 1


def test_struct_layouts_match_header():
    from spvipes_amd import _abi

    # spv_counts: ptr, i64, ptr, i32, i32  -> 32 bytes;  spv_dec_params: count the fields of the header
    assert ctypes.sizeof(_abi.SpvCounts) == 32
    src = open(os.path.join(ROOT, "include", "spvipes_hip.h")).read()
    body = re.search(r"typedef struct spv_dec_params \{(.*?)\} spv_dec_params;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    n_fields = 0
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            n_fields += decl.count(",") + 1
    assert n_fields == len(_abi.SpvDecParams._fields_)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spvipes_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in text.replace("# oracle", ""), f"{fn} mentions the oracle: the product path must not use it"
