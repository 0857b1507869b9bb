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
SYNTH_LIB = os.path.join(LIBDIR, "libndt_synth.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libndt_oracle.so")

HIP_SOURCES = ["ndt2d_api.hip"]
# -fno-slp-vectorize: at -O3 hipcc packs adjacent f32 ops into v_pk_*_f32, which gfx950 issues
# at half rate and which needs v_mov shuffles to form register pairs: measured 12 % slower on
# the loop-closure kernel.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
             "-Wno-unused-function", "-fno-slp-vectorize"]
# RCCL (the multi-device gather of ndt2d_multi_align_dev) and roctx (marker ranges for rocprofv3 --marker-trace) are
# loaded on first use with dlopen (csrc/ndt_dyn.hpp), not linked: the library loads on a machine without either
HIP_LIBS = ["-ldl"]


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
           *[os.path.join(CSRC, s) for s in HIP_SOURCES], *HIP_LIBS]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return LIB


def build_synth(force: bool = False, verbose: bool = False) -> str:
    """The device-side workload generator (include/ndt_synth.h): its own library, so the matcher
    library carries nothing of it.  -ffp-contract=off: it must reproduce synth.py bit for bit."""
    src = os.path.join(CSRC, "ndt_synth.hip")
    if not force and _newer(SYNTH_LIB, [src, os.path.join(ROOT, "include", "ndt_synth.h")]):
        return SYNTH_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "include"), "-o", SYNTH_LIB, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return SYNTH_LIB


def _cpu_tag() -> str:
    """Short hash of this host's CPU flags: a -march=native build must not travel to another CPU."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def build_oracle(force: bool = False, verbose: bool = False, native: bool = False) -> str | None:
    """gcc build of oracle/ndt_oracle.c (the C restatement; checker and timed CPU baseline).
    native=True: a second library built with -O3 -march=native for THIS host's CPU (file name
    carries a hash of the CPU flags), used by bench.py's cpu_baseline leg on the box it runs on."""
    src = os.path.join(ORACLE_DIR, "ndt_oracle.c")
    if not os.path.exists(src):
        return None
    out = ORACLE_LIB if not native else os.path.join(ORACLE_DIR, "_build", f"libndt_oracle_native_{_cpu_tag()}.so")
    if not force and _newer(out, [src]):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    opt = ["-O3", "-march=native"] if native else ["-O2"]
    cmd = ["gcc", *opt, "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-o", out, src, "-lm"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return out


def build_all(force: bool = False, verbose: bool = False):
    return build_hip(force, verbose), build_synth(force, verbose), build_oracle(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
