"""Submap persistence (ndt2d_save_map / ndt2d_load_map and the 3D twins): the cached grid leaves a handle as its exact
per-cell sums and comes back bit for bit - records, alignments and later submap updates included."""
import ctypes as C

import numpy as np
import pytest

from gtsam_ndt_amd import synth, synth3d

pytestmark = pytest.mark.gpu


def _same_grid(a, b):
    for x, y in zip(a.grid(), b.grid()):
        assert np.array_equal(x, y)
    ia, ib = a.grid_info(), b.grid_info()
    for f, _ in ia._fields_:
        assert getattr(ia, f) == getattr(ib, f), f


@pytest.mark.parametrize("overlap", [0, 4])
def test_2d_map_round_trip_is_bit_exact(gpu_lib, overlap):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=60_000, n_src=20_000)
    other = synth.make_pair(3, n_tgt=200_000, n_src=1000)               # a bigger grid the loading handle held before
    kw = dict(overlap_grids=overlap)
    with NdtMatcher2D(**kw) as a, NdtMatcher2D(**kw) as b, NdtMatcher2D(**kw) as small:
        a.set_target(d["tx"], d["ty"])
        blob = a.save_map()
        hdr = L.MapHeader.from_buffer_copy(blob[:C.sizeof(L.MapHeader)].tobytes())
        info = a.grid_info()
        assert (hdr.magic, hdr.version, hdr.dims, hdr.ngrid) == (L.MAP_MAGIC, 1, 2, 4 if overlap == 4 else 1)
        assert (hdr.width, hdr.height, hdr.depth, hdr.cell_bytes) == (info.width, info.height, 1, 48)
        assert blob.size == 104 + 48 * hdr.n_cells and hdr.n_points == 60_000 and hdr.cell_size == 0.5
        b.set_target(other["tx"], other["ty"])                          # stale records of another, larger map
        b.load_map(blob)
        small.load_map(blob.tobytes())                                  # a fresh handle, from bytes
        for m in (b, small):
            _same_grid(a, m)
            ra, rm = a.align(d["sx"], d["sy"], d["init"]), m.align(d["sx"], d["sy"], d["init"])
            assert ra.pose == rm.pose and np.array_equal(ra.H, rm.H) and ra.iterations == rm.iterations and ra.status == 0
        # the reloaded submap goes on taking points exactly as the original does
        x2, y2 = d["sx"], d["sy"]                                      # the scan, a few centimetres off: more points per cell
        na, nb = a.add_target_points(x2, y2), b.add_target_points(x2, y2)
        assert na == nb
        _same_grid(a, b)
        assert np.array_equal(a.save_map(), b.save_map())


def test_2d_map_reloads_under_other_validity_rules(gpu_lib):
    """The sums are parameter-free: loading with min_points = 12 gives the grid set_target builds with min_points = 12."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=40_000, n_src=1000)
    with NdtMatcher2D() as a, NdtMatcher2D(min_points=12, eig_ratio=0.01) as b, NdtMatcher2D(min_points=12, eig_ratio=0.01) as ref:
        a.set_target(d["tx"], d["ty"])
        b.load_map(a.save_map())
        ref.set_target(d["tx"], d["ty"])
        _same_grid(b, ref)
        assert b.grid_info().n_valid < a.grid_info().n_valid


def test_map_errors(gpu_lib):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtMatcher3D
    lib = L.load()
    d = synth.make_pair(1)
    d3 = synth3d.make_pair3d(n_elev=16, n_azim=256)
    with NdtMatcher2D() as a, NdtMatcher2D(cell_size=0.25) as fine, NdtMatcher2D(overlap_grids=4) as four, NdtMatcher3D() as m3:
        with pytest.raises(L.NdtError) as e:
            a.save_map()
        assert e.value.code == L.NDT_ERR_NO_TARGET and lib.ndt2d_map_size(a._h) == 0
        a.set_target(d["tx"], d["ty"])
        blob = a.save_map()
        need = C.c_size_t(0)
        short = np.empty(200, np.uint8)
        assert lib.ndt2d_save_map(a._h, short.ctypes.data, short.size, C.byref(need)) == L.NDT_ERR_CAPACITY and need.value == blob.size
        bad = blob.copy(); bad[0] ^= 0xFF
        for h, buf in ((a, bad), (a, blob[:-8]), (a, blob[:50]), (fine, blob), (four, blob)):
            with pytest.raises(L.NdtError) as e:
                h.load_map(buf)
            assert e.value.code == L.NDT_ERR_INVALID_ARG
        with pytest.raises(L.NdtError) as e:
            m3.load_map(blob)                                            # a 2D map into a 3D handle
        assert e.value.code == L.NDT_ERR_INVALID_ARG
        m3.set_target(d3["tx"], d3["ty"], d3["tz"])
        with pytest.raises(L.NdtError):
            a.load_map(m3.save_map())
        # a failed load leaves the handle without a target rather than with half of one
        r = a.align(d["sx"], d["sy"], d["init"])
        assert r.status == 0                                             # header errors are caught before anything is touched


def test_3d_map_round_trip_is_bit_exact(gpu_lib):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d = synth3d.make_pair3d(n_elev=32, n_azim=1024)
    big = synth3d.make_pair3d(n_elev=64, n_azim=512, pose=(0.3, 0.1, 0.0, 0.0, 0.0, 0.1))
    for mode in (0, 1):
        with NdtMatcher3D(hessian_mode=mode) as a, NdtMatcher3D(hessian_mode=mode, cell_size=1.0) as b:
            a.set_target(d["tx"], d["ty"], d["tz"])
            blob = a.save_map()
            hdr = L.MapHeader.from_buffer_copy(blob[:104].tobytes())
            info = a.grid_info()
            assert (hdr.dims, hdr.ngrid, hdr.cell_bytes) == (3, 1, 80)
            assert (hdr.width, hdr.height, hdr.depth) == (info.width, info.height, info.depth) and blob.size == 104 + 80 * hdr.n_cells
            b.set_target(big["tx"], big["ty"], big["tz"])
            b.load_map(blob)
            _same_grid(a, b)
            ra, rb = a.align(d["sx"], d["sy"], d["sz"], d["init"]), b.align(d["sx"], d["sy"], d["sz"], d["init"])
            assert ra.pose == rb.pose and np.array_equal(ra.H, rb.H) and (ra.status, ra.iterations) == (rb.status, rb.iterations)
            assert mode == 1 or ra.status == 0                           # (Newton steps on this sparse pair wander; the contract is equality)
            na, nb = a.add_target_points(big["tx"], big["ty"], big["tz"]), b.add_target_points(big["tx"], big["ty"], big["tz"])
            assert na == nb
            _same_grid(a, b)
            assert np.array_equal(a.save_map(), b.save_map())


def test_2d_load_keeps_the_outer_ring_empty(gpu_lib):
    """Alignments clamp out-of-range lookups onto the grid's outermost ring, which the builders never fill: a map
    edited by hand (or corrupted) with points in ring cells loads with those cells cleared."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(1)
    with NdtMatcher2D() as a, NdtMatcher2D() as b:
        a.set_target(d["tx"], d["ty"])
        blob = a.save_map().copy()
        hdr = L.MapHeader.from_buffer_copy(blob[:104].tobytes())
        cells = blob[104:].view(np.int64).reshape(-1, 6)              # sx sy sxx sxy syy (n | pad)
        inner = int(np.argmax(cells[:, 5] & 0xFFFFFFFF))              # the fullest cell's sums ...
        for ring_cell in (0, hdr.width - 1, hdr.width * (hdr.height - 1) + 3, 2 * hdr.width):    # ... copied onto the ring
            cells[ring_cell] = cells[inner]
        b.load_map(blob)
        for x, y in zip(a.grid(), b.grid()):
            assert np.array_equal(x, y)                                # the ring is empty again, nothing else changed
        assert b.grid_info().n_valid == a.grid_info().n_valid


def test_forged_maps_are_refused(gpu_lib):
    """A buffer that did not come from ndt*_save_map: a 2D header with depth > 1, a non-finite origin, a
    cell count beyond the capacity the sums are exact for, sums no point set can have (negative sum of squares, a
    variance numerator below zero, coordinates outside the cell) - each is refused with NDT_ERR_INVALID_ARG before
    anything is uploaded, and the handle keeps working afterwards; the untouched blob still loads."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtMatcher3D
    d = synth.make_pair(1)
    d3 = synth3d.make_pair3d(n_elev=16, n_azim=256)

    def refused(h, buf):
        with pytest.raises(L.NdtError) as e:
            h.load_map(buf)
        assert e.value.code == L.NDT_ERR_INVALID_ARG

    with NdtMatcher2D() as a, NdtMatcher2D() as b:
        a.set_target(d["tx"], d["ty"])
        blob = a.save_map().copy()
        hdr = L.MapHeader.from_buffer_copy(blob[:104].tobytes())
        off = {name: getattr(L.MapHeader, name).offset for name, _ in L.MapHeader._fields_}

        def with_header(**kw):
            out = blob.copy()
            h2 = L.MapHeader.from_buffer_copy(blob[:104].tobytes())
            for k, v in kw.items():
                setattr(h2, k, v)
            out[:104] = np.frombuffer(bytes(h2), dtype=np.uint8)
            return out

        # depth 2 with the cell count to match (and a buffer long enough): still not a 2D map
        deep = np.concatenate([with_header(depth=2, n_cells=2 * hdr.n_cells), blob[104:]])
        refused(b, deep)
        for bad_value in (np.float32(np.nan), np.float32(np.inf)):
            nanmap = blob.copy()
            nanmap[off["origin"]:off["origin"] + 4] = np.frombuffer(bad_value.tobytes(), dtype=np.uint8)
            refused(b, nanmap)
        cells_of = lambda buf: buf[104:].view(np.int64).reshape(-1, 6)       # sx sy sxx sxy syy (n | pad)
        k = int(np.argmax(cells_of(blob)[:, 5] & 0xFFFFFFFF))
        for col, value in ((5, (1 << 20) + 1), (2, -5), (0, 1 << 62), (3, -(1 << 62))):
            forged = blob.copy()
            cells_of(forged)[k, col] = value
            refused(b, forged)
        forged = blob.copy()                                                   # n * sxx < sx^2: no such point set
        c = cells_of(forged)[k]
        c[2] = (int(c[0]) * int(c[0])) // int(c[5] & 0xFFFFFFFF) - 10
        refused(b, forged)
        b.load_map(blob)                                                       # the real thing still loads ...
        assert a.align(d["sx"], d["sy"], d["init"]).pose == b.align(d["sx"], d["sy"], d["init"]).pose
    with NdtMatcher3D() as a3, NdtMatcher3D() as b3:
        a3.set_target(d3["tx"], d3["ty"], d3["tz"])
        blob3 = a3.save_map().copy()
        cells3 = lambda buf: buf[104:].view(np.int64).reshape(-1, 10)          # s[3] ss[6] (n | pad)
        k = int(np.argmax(cells3(blob3)[:, 9] & 0xFFFFFFFF))
        for col, value in ((9, (1 << 20) + 7), (3, -1), (1, -(1 << 61)), (7, 1 << 62)):
            forged = blob3.copy()
            cells3(forged)[k, col] = value
            refused(b3, forged)
        b3.load_map(blob3)
        assert a3.align(d3["sx"], d3["sy"], d3["sz"], d3["init"]).pose == b3.align(d3["sx"], d3["sy"], d3["sz"], d3["init"]).pose
