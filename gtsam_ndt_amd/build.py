"""In-tree builds: the gfx950 C-ABI library (hipcc) and, for tests/bench only, the C oracle.

Nothing here is a JIT: artefacts land next to the sources (gtsam_ndt_amd/lib/, oracle/_build/)
so that they travel to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gtsam_ndt_amd", "csrc")
LIBDIR = os.path.join(ROOT, "gtsam_ndt_amd", "lib")
LIB = os.path.join(LIBDIR, "libndt_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libndt_oracle.so")

HIP_SOURCES = ["ndt2d_api.hip"]
# -fno-slp-vectorize: at -O3 hipcc packs adjacent f32 ops into v_pk_*_f32, which gfx950 issues
# at half rate and which needs v_mov shuffles to form register pairs: measured 12 % slower on
# the loop-closure kernel.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
             "-Wno-unused-function", "-fno-slp-vectorize"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def _deps(dirs: list[str]) -> list[str]:
    out = []
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".hip", ".hpp", ".h", ".cpp", ".c")):
                out.append(os.path.join(d, f))
    return out


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build the gfx950 NDT library")
    return p


def build_hip(force: bool = False, verbose: bool = False) -> str:
    deps = _deps([CSRC, os.path.join(ROOT, "include")])
    if not force and _newer(LIB, deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc_path(), *HIP_FLAGS, "-I", os.path.join(ROOT, "include"), "-o", LIB,
           *[os.path.join(CSRC, s) for s in HIP_SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return LIB


def build_oracle(force: bool = False, verbose: bool = False) -> str | None:
    """gcc build of oracle/ndt_oracle.c (the C restatement used as the timed CPU baseline)."""
    src = os.path.join(ORACLE_DIR, "ndt_oracle.c")
    if not os.path.exists(src):
        return None
    if not force and _newer(ORACLE_LIB, [src]):
        return ORACLE_LIB
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-o", ORACLE_LIB,
           src, "-lm"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return ORACLE_LIB


def build_all(force: bool = False, verbose: bool = False):
    return build_hip(force, verbose), build_oracle(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
