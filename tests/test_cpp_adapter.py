"""The header-only C++ adapter (include/ndt_matcher_hip.hpp): compiles with plain g++ against
the C ABI; without a GPU it must fail loudly with NDT_ERR_NO_DEVICE; on the GPU box it must
reproduce the oracle's pose."""
import os
import subprocess

import numpy as np
import pytest

from gtsam_ndt_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory, ndt_lib):
    out = tmp_path_factory.mktemp("cpp") / "adapter_smoke"
    libdir = os.path.join(ROOT, "gtsam_ndt_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"), "-o", str(out),
                    "-L", libdir, "-lndt_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                    "-lamdhip64"],
                   check=True)
    return str(out)


def _run(exe, tmp_path):
    d = synth.make_pair(1)
    paths = []
    for k in ("tx", "ty", "sx", "sy"):
        p = tmp_path / f"{k}.f32"
        d[k].tofile(p)
        paths.append(str(p))
    r = subprocess.run([exe, *paths, *[repr(v) for v in d["init"]]], capture_output=True, text=True, timeout=120)
    return d, r


def test_adapter_compiles_and_fails_loudly_without_gpu(exe, tmp_path, ndt_lib):
    if ndt_lib.ndt_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    _, r = _run(exe, tmp_path)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


def test_gtsam_header_is_inert_without_gtsam(tmp_path):
    src = tmp_path / "g.cpp"
    src.write_text('#include "ndt_gtsam_factor.hpp"\nint main(){ndt::Pose2 p; (void)p; return 0;}\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "g.o")], check=True)


@pytest.mark.gpu
def test_adapter_matches_oracle(exe, tmp_path, gpu_lib):
    from oracle import ndt2d as o
    d, r = _run(exe, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict((l.split()[0], l.split()[1:]) for l in r.stdout.strip().splitlines())
    prm = o.NdtParams()
    ref = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    for key in ("single", "batch", "multi"):
        pose = np.array([float(v) for v in lines[key][:3]])
        assert np.abs(pose - np.array(ref["pose"])).max() < 1e-4
        assert int(lines[key][-1]) == 0
    assert lines["multi"] == lines["batch"]          # same kernel, same pair: bit-identical
    assert lines["maprt"] == ["1"]                  # saveMap -> loadMap into a second matcher: the same alignment
    # device-pointer forms through the C++ adapter: the resident scan, several starts / scans per chain (start 0
    # and scan 0 are the plain alignment: bit-identical to it), the RCCL gather of the multi-device context
    assert lines["dev"][:5] == lines["single"][:3] + [lines["single"][3], lines["single"][5]]
    one = np.array([float(v) for v in lines["single"][:3]])      # (1000-point scan: the single call runs the one-workgroup
    for key in ("multistart", "multiscan"):                       #  kernel, the chains k_iterate's order: equal to rounding)
        assert np.abs(np.array([float(v) for v in lines[key][:3]]) - one).max() < 2e-6 and int(lines[key][4]) == 0
    assert lines["multistart"][5] == "3"
    assert lines["rccl"] == lines["batch"]
    # the coarse-to-fine batch through the adapter = the same call through the ctypes binding
    from gtsam_ndt_amd.matcher import NdtBatch2D, pyramid_params
    guess = (d["init"][0] + 0.5, d["init"][1] - 0.4, d["init"][2] + 0.04)
    with NdtBatch2D(levels=pyramid_params()) as bp:
        rp = bp.align([(d["tx"], d["ty"])], [(d["sx"], d["sy"])], [guess])[0]
    got = [float(v) for v in lines["pyramid"][:3]]
    assert got == list(rp.pose) and int(lines["pyramid"][3]) == rp.iterations and int(lines["pyramid"][4]) == rp.status
    # Biber's four overlapping grids through the adapter: single-pair path and batch path agree
    ov = [float(v) for v in lines["overlap"]]
    assert int(ov[4]) == 0 and int(ov[9]) == 0 and max(abs(ov[k] - ov[5 + k]) for k in range(3)) < 2e-5
    prm4 = o.NdtParams(overlap=4)
    ref4 = o.align(o.build_grids(d["tx"], d["ty"], prm4), d["sx"], d["sy"], d["init"], prm4)
    assert max(abs(ov[k] - ref4["pose"][k]) for k in range(3)) < 1e-4
    assert abs(float(lines["infocov"][0]) - 1.0) < 1e-6
    assert abs(float(lines["localcov"][0]) - 1.0) < 1e-9 and abs(float(lines["localcov"][1]) - 1.0) < 1e-12
    # the 3D adapter on a self-generated room: the known motion (0.20, -0.15, 0.05, yaw 0.02) comes back
    p3 = np.array([float(v) for v in lines["three_d"][:6]])
    assert int(lines["three_d"][7]) in (0, 1)
    assert np.abs(p3 - np.array([0.20, -0.15, 0.05, 0.0, 0.0, 0.02])).max() < 2e-2, p3
    assert abs(float(lines["three_d"][8]) - 1.0) < 1e-6
    assert float(lines["localcov3"][0]) < 1e-4          # J d d' J' equals the finite step's tangent vector squared to first order
    # ... and through the 3D batch (NdtBatchHip3), from the identity and from a displaced guess
    b0, b1 = (np.array([float(v) for v in lines[k][:6]]) for k in ("three_d_batch0", "three_d_batch1"))
    assert int(lines["three_d_batch0"][7]) in (0, 1) and int(lines["three_d_batch1"][7]) in (0, 1)
    assert np.abs(b0 - np.array([0.20, -0.15, 0.05, 0.0, 0.0, 0.02])).max() < 2e-2 and np.abs(b1 - b0).max() < 2e-3, (b0, b1)
