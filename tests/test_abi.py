"""The C-ABI library loads without a GPU, exports every symbol include/ndt_hip.h declares,
its POD structs have the layout the ctypes mirror assumes, and it fails loudly (no CPU
fallback) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ndt_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ndt[0-9a-z_]*)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(ndt_lib):
    from gtsam_ndt_amd import _lib
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(ndt_lib, n), f"{n} declared in ndt_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES and ndt_hip.h disagree"
    assert ndt_lib.ndt_abi_version() == 1
    assert ndt_lib.ndt_status_string(0) == b"ok"
    assert b"degenerate" in ndt_lib.ndt_status_string(2)


def test_struct_layout_matches_the_header(tmp_path):
    from gtsam_ndt_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ndt_hip.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ndt2d_params), sizeof(ndt2d_result),'
                    'sizeof(ndt2d_eval), sizeof(ndt2d_grid_info), offsetof(ndt2d_params, eig_ratio),'
                    'offsetof(ndt2d_result, score), offsetof(ndt2d_params, min_hits), sizeof(ndt_map_header),'
                    'offsetof(ndt_map_header, cell_size), offsetof(ndt_map_header, origin));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out == [C.sizeof(_lib.Params2D), C.sizeof(_lib.Result2D), C.sizeof(_lib.Eval2D),
                   C.sizeof(_lib.GridInfo2D), _lib.Params2D.eig_ratio.offset, _lib.Result2D.score.offset,
                   _lib.Params2D.min_hits.offset, C.sizeof(_lib.MapHeader), _lib.MapHeader.cell_size.offset,
                   _lib.MapHeader.origin.offset]
    assert C.sizeof(_lib.MapHeader) == 104


def test_header_is_plain_c(tmp_path):
    prog = tmp_path / "c.c"
    prog.write_text('#include "ndt_hip.h"\nint main(void){ndt2d_params p; (void)p; return NDT_OK;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-c", str(prog), "-o", str(tmp_path / "c.o")], check=True)


def test_default_params_and_validation(ndt_lib):
    from gtsam_ndt_amd import _lib
    p = _lib.Params2D()
    ndt_lib.ndt2d_default_params(C.byref(p))
    assert (p.cell_size, p.min_points, p.eig_ratio, p.d1, p.d2) == (0.5, 3, 1e-3, 1.0, 1.0)
    assert p.hessian_mode == 0 and p.max_iterations == 100 and p.fixed_iterations == 0
    h = C.c_void_p()
    if ndt_lib.ndt_device_count() > 0:
        pytest.skip("validation-without-device check is for the CPU container")
    # no device: creation must fail loudly, never fall back to a CPU path
    assert ndt_lib.ndt2d_create(C.byref(p), 0, C.byref(h)) == _lib.NDT_ERR_NO_DEVICE
    assert not h.value
    p.cell_size = -1.0
    assert ndt_lib.ndt2d_create(C.byref(p), 0, C.byref(h)) == _lib.NDT_ERR_INVALID_ARG
    assert ndt_lib.ndt2d_create(None, 0, C.byref(h)) == _lib.NDT_ERR_INVALID_ARG


def test_matcher_raises_without_device(ndt_lib):
    from gtsam_ndt_amd import _lib
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    if ndt_lib.ndt_device_count() > 0:
        pytest.skip("needs the GPU-less container")
    with pytest.raises(_lib.NdtError) as e:
        NdtMatcher2D()
    assert e.value.code == _lib.NDT_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    # every context type: the batch, multi-device, 3D and 3D-batch entry points fail the same way
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtBatch3D, NdtMatcher3D, NdtMulti2D
    for make in (NdtBatch2D, NdtBatch3D, NdtMatcher3D, lambda: NdtMulti2D(devices=[0])):
        with pytest.raises(_lib.NdtError) as e:
            make()
        assert e.value.code == _lib.NDT_ERR_NO_DEVICE, make


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under gtsam_ndt_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "gtsam_ndt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if f == "build.py":
                    continue      # builds the checker (allowed), never calls it
                assert "from oracle" not in txt and "import oracle" not in txt and "ndt_oracle" not in txt, f
