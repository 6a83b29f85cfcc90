"""Builds libspvipes_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m spvipes_amd.build          # or: from spvipes_amd.build import build; build()

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libspvipes_hip.so")
SOURCES = ["spv_abi.hip"]
HEADERS = ["spv_common.h", "spv_gemm.h", "spv_decoder.h", "spv_small.h", os.path.join("..", "..", "include", "spvipes_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7 with gfx950 support)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]   # (the flags live in this file)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
        # No SLP vectorisation: the packed-fp32 (v_pk_*_f32 with op_sel) code it forms in the likelihood kernel gave results
        # that were not bit-reproducible from run to run on gfx950 / ROCm 7.2 (one gradient element of the last 16 lanes of
        # a wave, about 1 launch in 500 under GPU sharing).  Costs about 1 % of the step.
        "-fno-slp-vectorize",
        *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp",
    ]
    if verbose:
        print("[spvipes_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
