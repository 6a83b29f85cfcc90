"""CPU: static checks on the gfx950 code object inside libspvipes_hip.so (disassembled with llvm-objdump; no GPU needed).

Pins what DESIGN.md section 8 relies on: the NB-mixture likelihood kernel's chunk loop contains NO packed-fp32 instruction
(v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  With LLVM's SLP vectoriser on, that loop's gradient arithmetic is packed into
v_pk_* operations with op_sel / op_sel_hi operand selection, and round 1 measured gradients that were not bit-reproducible
from run to run under GPU sharing with exactly that code (cause not established); the library is therefore built with
-fno-slp-vectorize, and this test fails if a compiler upgrade, a flag change or a source edit brings such instructions back.
The few plain v_pk_add_f32 the LOOP vectoriser forms in the kernel's epilogue (the LDS reduction of d theta, after the last
MFMA, no operand-select modifiers) are allowed."""
import os
import re
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    from spvipes_amd import _abi
    from spvipes_amd.build import build

    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    build()
    d = tmp_path_factory.mktemp("isa")
    so = shutil.copy(_abi.LIB_PATH, d / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", str(so)], check=True, capture_output=True, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert co, "no gfx950 code object inside the library"
    out = subprocess.run([OBJDUMP, "-d", str(d / co[0])], check=True, capture_output=True, text=True).stdout
    funcs, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name is not None and line.startswith("\t"):
            funcs[name].append(line.strip().split("//")[0].strip())
    return funcs


def test_likelihood_kernel_loop_has_no_packed_fp32(disassembly):
    kernels = {k: v for k, v in disassembly.items() if "13dec_nb_kernelILb1" in k}   # the TRAIN instantiations
    assert len(kernels) >= 6, [k for k in disassembly if "dec_nb" in k]   # {bf16, split} x {f16, f32 logits} x count modes
    pk = re.compile(r"^v_pk_(mul|fma|add)_f32\b")
    for name, ins in kernels.items():
        last_mfma = max(i for i, s in enumerate(ins) if s.startswith("v_mfma"))
        packed = [(i, s) for i, s in enumerate(ins) if pk.match(s)]
        in_loop = [s for i, s in packed if i < last_mfma]
        assert not in_loop, f"{name}: packed fp32 inside the chunk loop: {in_loop[:3]}"
        modified = [s for _, s in packed if "op_sel" in s or "neg_" in s]
        assert not modified, f"{name}: packed fp32 with operand-select modifiers: {modified[:3]}"
        assert len(packed) <= 24, f"{name}: {len(packed)} packed fp32 instructions (expected only the epilogue's d-theta reduction)"


def test_dma_kernels_keep_their_queue_in_flight(disassembly):
    """The multi-stage LDS-DMA GEMMs (csrc/spv_fc1.h, the four-stage kernel of csrc/spv_dec_gemm.h) rely on a COUNTED s_waitcnt vmcnt(N) at the top of every K tile: a compiler-inserted
    vmcnt(0) in front of the fragment reads (what hipcc emits for the transposed-read builtin) would drain the two tiles in
    flight once per tile."""
    for key in ("fc1_fwd_dma_kernel", "fc1_fwd_dma_pair_kernel", "fc1_wgrad_dma_kernel", "fc1_wgrad_dma_pair_kernel", "fc1_wgrad_dma_wide_pair_kernel", "dec_gemm320_dma4_kernel"):
        ks = {k: v for k, v in disassembly.items() if key in k}
        assert ks, key
        for name, ins in ks.items():
            assert any("global_load_lds_dwordx4" in s for s in ins), name
            counted = [s for s in ins if s.startswith("s_waitcnt") and re.search(r"vmcnt\([1-9]", s)]
            assert counted, f"{name}: no counted vmcnt wait"
            # between a tile's barrier and its first MFMA (where the fragment reads sit) nothing may wait on the DMA queue
            after_barrier = False
            for i_, s_ in enumerate(ins):
                if s_.startswith("s_barrier") and any(re.search(r"vmcnt\([1-9]", x) for x in ins[max(0, i_ - 4):i_]):   # the K loop's barrier
                    after_barrier = True
                elif s_.startswith("v_mfma") or s_.startswith("global_store") or s_.startswith("global_atomic"):
                    after_barrier = False   # MFMAs reached; or (listing order: the loop's back edge sits above) the epilogue began -- the split-K fix-up there drains on purpose
                elif after_barrier and s_.startswith("s_waitcnt") and "vmcnt" in s_:
                    raise AssertionError(f"{name}: '{s_}' between the tile barrier and the MFMAs drains the DMA pipeline")
