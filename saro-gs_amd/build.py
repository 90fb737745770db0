"""Builds the gfx950 shared library (C ABI of include/gsrast.h) in-tree with hipcc.

The ROCm counterpart of the reference's setup.py (submodules/gaussian_rasterization_ch3/setup.py:17-33),
except that the product is a plain C-ABI .so loaded with ctypes, not a torch extension: no torch
headers are involved and the library has no Python dependency.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "diff_gaussian_rasterization_ch3", "libgsrast_hip.so")
SOURCES = ["gsrast_capi.hip"]
HEADERS = ["gsrast_common.h", "gsrast_policy.h", "gsrast_preprocess.h", "gsrast_binning.h", "gsrast_blend.h", "gsrast_loss.h", "gsrast_epilogue.h", "gsrast_adam.h", "gsrast_knn.h", "gsrast_hexplane.h", "gsrast_exchange.h",
           os.path.join("..", "..", "include", "gsrast.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",                            # FMAs only where written: bit parity with the oracle
    "-fhip-fp32-correctly-rounded-divide-sqrt",     # IEEE divide / sqrt in the per-Gaussian kernels
    "-munsafe-fp-atomics",                          # hardware global_atomic_add_f32, no CAS loop
    "-fno-slp-vectorize",                           # packed fp32 VALU is half rate on gfx950 (tools/valu_calib.hip): SLP-formed v_pk_* plus the
                                                    # v_mov shuffles that feed them cost more than the scalar ops (blend_bwd -5 %)
    "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC_DIR, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + [os.path.join(SRC_DIR, f) for f in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
