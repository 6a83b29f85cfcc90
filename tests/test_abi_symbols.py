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


def test_every_ctypes_struct_has_the_size_the_c_header_gives(tmp_path):
    """Compile a few lines of C against include/spvipes_hip.h (plain gcc, no GPU) and compare sizeof() of every struct
    with its ctypes mirror."""
    import subprocess

    from spvipes_amd import _abi

    pairs = {
        "spv_counts": _abi.SpvCounts, "spv_dec_params": _abi.SpvDecParams, "spv_linear_prob": _abi.SpvLinearProb,
        "spv_linear_batch": _abi.SpvLinearBatch, "spv_bn_prob": _abi.SpvBnProb, "spv_bn_batch": _abi.SpvBnBatch,
        "spv_sample_prob": _abi.SpvSampleProb, "spv_sample_batch": _abi.SpvSampleBatch, "spv_poe_args": _abi.SpvPoeArgs,
        "spv_zsplit_args": _abi.SpvZsplitArgs, "spv_fold_prob": _abi.SpvFoldProb, "spv_fold_batch": _abi.SpvFoldBatch,
        "spv_reduce_prob": _abi.SpvReduceProb, "spv_reduce_batch": _abi.SpvReduceBatch, "spv_plan": _abi.SpvPlan,
        "spv_plan_expert_args": _abi.SpvPlanExpertArgs, "spv_poe_comp_args": _abi.SpvPoeCompArgs,
        "spv_trunk_prob": _abi.SpvTrunkProb, "spv_trunk_batch": _abi.SpvTrunkBatch, "spv_gemm_fixup": _abi.SpvGemmFixup, "spv_gemm_args": _abi.SpvGemmArgs, "spv_dec_group": _abi.SpvDecGroup,
    }
    src = tmp_path / "sizes.c"
    lines = ['#include <stdio.h>', f'#include "{os.path.join(ROOT, "include", "spvipes_hip.h")}"', "int main(void) {"]
    lines += [f'  printf("{n} %zu\\n", sizeof({n}));' for n in pairs]
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for name, cls in pairs.items():
        assert int(got[name]) == ctypes.sizeof(cls), f"{name}: C header {got[name]} bytes, ctypes {ctypes.sizeof(cls)}"
