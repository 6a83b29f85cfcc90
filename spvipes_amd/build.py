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
HEADERS = ["spv_common.h", "spv_gemm.h", "spv_fc1.h", "spv_dec_gemm.h", "spv_decoder.h", "spv_small.h", "spv_poe_n.h", os.path.join("..", "..", "include", "spvipes_hip.h")]


FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
    # No SLP vectorisation: see DESIGN.md section 8 (run-to-run reproducibility of the likelihood kernel's gradients) and
    # tests/test_isa_invariants.py, which pins what the built code object may contain.
    "-fno-slp-vectorize",
]


def build_id() -> str:
    """Fingerprint of everything the library is built from: every source / header byte plus the compiler flags.  It is
    compiled into the library (``spv_build_id()``) and ``_abi.load()`` refuses a library whose fingerprint differs from
    the sources next to it -- a .so built from other sources or with other flags never loads silently."""
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.replace(os.sep, "/").encode() + b"\0" + f.read())
    return h.hexdigest()[:32]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7 with gfx950 support)")


def lib_build_id(path: str = LIB) -> str:
    """the fingerprint compiled into an existing library, read from the file without loading it ('' if absent)"""
    import re

    if not os.path.exists(path):
        return ""
    with open(path, "rb") as f:
        m = re.search(rb"SPV_BUILD_ID=([0-9a-f]{32})", f.read())
    return m.group(1).decode() if m else ""


def is_stale() -> bool:
    return lib_build_id() != build_id()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [
        _hipcc(), *FLAGS, f'-DSPV_BUILD_ID="{build_id()}"',
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
