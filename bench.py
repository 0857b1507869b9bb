#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X NDT scan matcher (driver contract).

    python bench.py --gpus N --steps K --warmup W           # any N: for N > 1 this process starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Metric (BASELINE.json): NDT Gauss-Newton iterations/sec @ 1M-point target, plus the
converged-pose error against the CPU reference (here: this repo's oracle - the reference
implementation is unavailable, /root/reference/README.md:1).

N = 1  value = BASELINE config 3: one pair, 1M-point submap target vs 100k-point scan.
       A step = one alignment of fixed K_GN = 30 Gauss-Newton iterations with the target grid
       cached and all inputs resident in HBM (the grid build is timed separately and
       reported as grid_build).  value = steps * 30 / elapsed.  The line also carries, under
       `batch`, the single-GPU figure of BASELINE config 4 (512 candidate pairs x 100k points).
N > 1  one process per GPU (RCCL).  A single alignment does not shard, so
       `value` is the N = 1 workload as one independent replica per GPU ("value_kind": "replicas":
       linear by construction, it exercises no multi-GPU code).  The path that DOES shard is in
       the same line under `batch`: config 4, 512 pairs per rank (4096 at N = 8), pairs split
       over the ranks with no data-path collective and ONE RCCL all_gather of the result rows
       inside every timed step.  batch.value = pair-iterations of all ranks / max-over-ranks
       time: read the 1 -> 8 scaling of the multi-GPU path from batch.value against the N = 1
       line's batch.value.

Started without WORLD_SIZE in the environment and with --gpus N > 1, this script launches
`python -m torch.distributed.run --nproc-per-node N` on itself as a CHILD process before any
GPU call, relays rank 0's JSON line and exits with the child's code (a process that has
touched the GPU is never re-executed).

One JSON line is printed by rank 0.  roofline / cpu_baseline are described in DESIGN.md section 7.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

K_GN = 30                      # fixed Gauss-Newton iterations per alignment in timed runs
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_POINT_ITER = 32      # 8 B point + 24 B cell record
METRIC = "NDT Gauss-Newton iters/sec @1M-pt target; converged-pose err vs CPU ref"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs-per-rank", type=int, default=512)
    ap.add_argument("--batch-points", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="N=1: skip the extra loop-closure batch figure")
    ap.add_argument("--no-batch-4096", action="store_true", help="N=1: skip the full 4096-pair single-GPU batch figure")
    ap.add_argument("--headline-only", action="store_true",
                    help="N=1, for profiling: only the timed config-3 steps, the config-4 batch (512 pairs) and config 5, without "
                         "the parity legs (converged alignment vs the oracle, batch cross-check) - a profile's kernel "
                         "populations are then exactly the timed fixed-K launches")
    ap.add_argument("--no-latency", action="store_true",
                    help="N=1: skip the converged-mode call latency (keeps a profile's k_iterate population to the timed steps)")
    ap.add_argument("--all-configs", action="store_true",
                    help="N=1: also time config 5 (3D) and a lidar-sized loop-closure batch for the per-config table")
    ap.add_argument("--host-path", action="store_true",
                    help="N=1: also time the host-pointer entry points (PCIe-inclusive; reported beside value)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--with-3d", action="store_true", help="(default now) kept for old command lines")
    ap.add_argument("--multi-starts", type=str, default="8,16,64", help="N=1: start counts of the multi-start figures")
    ap.add_argument("--no-3d", action="store_true", help="N=1: skip BASELINE config 5 (3D SE(3), 131072-point pair)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N>1 flow on a 1-GPU box: every rank uses cuda:0 and the result "
                         "gather runs over gloo on host copies (RCCL refuses two ranks on one device)")
    ap.add_argument("--converged-batch", action="store_true",
                    help="also time the loop-closure batch in converged mode (each pair runs until its own convergence test "
                         "passes) with STRIDED sharding at N > 1 (pair k -> rank k mod N: evens out iteration-count variance, "
                         "SURVEY 8e); reported under batch_converged")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N>1 code path (process group + all_gather) even with one rank")
    a = ap.parse_args()
    if a.headline_only:
        a.no_latency = True
        a.no_batch_4096 = True
    return a


def hip_events_ms(stream_ptr: int, fn):
    """Time fn() with HIP events recorded on the stream the kernels are launched on."""
    s = torch.cuda.ExternalStream(stream_ptr)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s)
    fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1)


def stream_copy_GBps(dev, nbytes: int = 1 << 30, reps: int = 5) -> float:
    """On-box streaming ceiling (SURVEY.md 8d asks for it beside the 8 TB/s spec): a device-to-
    device copy of `nbytes`, counted as read + write traffic, best of `reps`."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dst.copy_(src); e1.record()
        e1.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def single_pair_rate(dev, dev_index, cfg: int, steps: int, warmup: int, n_src: int | None = None):
    """Fixed-K alignments of one synthetic pair of BASELINE config `cfg` (1 or 2), as the
    config-3 headline is timed: grid cached, inputs in HBM, HIP events over the timed region."""
    from gtsam_ndt_amd import synth
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(cfg) if n_src is None else synth.make_pair(cfg, n_src=n_src)
    tx, ty, sx, sy = (torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "sx", "sy"))
    with NdtMatcher2D(device=dev_index, fixed_iterations=K_GN) as m:
        m.set_target(tx, ty)
        torch.cuda.synchronize()
        for _ in range(warmup):
            m.align_async(sx, sy, d["init"], producer_complete=True)
        m.finish()
        t0 = time.perf_counter()
        ev_ms = hip_events_ms(m.stream, lambda: [m.align_async(sx, sy, d["init"], producer_complete=True) for _ in range(steps)])
        m.finish()
        el = time.perf_counter() - t0
    n_src = int(sx.numel())
    short = n_src <= 4096          # one workgroup runs the whole loop (k_align_small): one launch per alignment
    iter_us = 1e3 * ev_ms / (steps * (K_GN if short else K_GN + 1))
    alg = n_src * BYTES_PER_POINT_ITER
    return {"config": cfg, "n_target": int(tx.numel()), "n_source": n_src,
            "kernel": "k_align_small (whole loop in one workgroup)" if short else "k_iterate (one launch per iteration)",
            "iters_per_s": round(steps * K_GN / el, 1), "us_per_iteration": round(iter_us, 3),
            "algorithmic_GBps": round(alg / (iter_us * 1e-6) / 1e9, 1),
            "frac_of_8TBps": round(alg / (iter_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)}


def build_legs(m, tx, ty, sx, sy, pose, reps: int = 30):
    """The step in front of the path: ndt2d_set_target_dev of the 1M-point submap and
    ndt2d_add_target_points_dev of the 100k-point scan moved into the map frame by a pose (a front end pays one of them per
    scan).  Host call to return, and the GPU-side span by HIP events on the handle's stream (first kernel's start to the end
    of the one-wave kernel that publishes the counters to pinned host memory: kernels + launch boundaries).  Roofline numerator = SURVEY 8d's B_grid:
    8 B per point + 24 B per cell of the grid (for the update: the cells of the tiles the scan touches are not known to
    the host, so its figure counts the points only)."""
    st = torch.cuda.ExternalStream(m.stream)

    def timed(fn):
        host, span = [], []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(st); fn(); e1.record(st)
            host.append(time.perf_counter() - t0)
            e1.synchronize()
            span.append(e0.elapsed_time(e1))
        return 1e3 * float(np.median(host[3:])), float(np.median(span[3:]))

    info = m.set_target(tx, ty)
    h_ms, g_ms = timed(lambda: m.set_target(tx, ty))
    n_t, cells = int(tx.numel()), int(info.width) * int(info.height)
    b_grid = 8 * n_t + 24 * cells
    grid = {"workload": f"ndt2d_set_target_dev: {n_t} points -> {info.width} x {info.height} cells of 0.5 m ({info.n_valid} valid)",
            "ms_per_call": round(h_ms, 4), "gpu_span_us": round(1e3 * g_ms, 2),
            "kernels": "k_bounds_parts -> k_chunk_sort (geometry in its prologue) -> k_tile_gather",
            "roofline": {"bound": "hbm", "algorithmic_bytes": b_grid, "bytes_rule": "SURVEY 8d B_grid = 8 B x points + 24 B x cells",
                         "achieved": round(b_grid / (g_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(b_grid / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "timing": "HIP events on the handle's stream around the call (three kernels, their boundaries and the one-wave publish kernel)"}}
    h_ms, g_ms = timed(lambda: m.add_target_points(sx, sy, pose=pose))
    n_s = int(sx.numel())
    upd = {"workload": f"ndt2d_add_target_points_dev: a {n_s}-point scan moved by a pose and merged into the {n_t}-point submap "
                       "(exact: the grid equals a rebuild from all points)",
           "ms_per_call": round(h_ms, 4), "gpu_span_us": round(1e3 * g_ms, 2), "kernels": "k_chunk_sort (motion fused) -> k_tile_gather",
           "algorithmic_GBps": round(8 * n_s / (g_ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(8 * n_s / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    return grid, upd


def ordered_build_leg(dev, dev_index, reps: int = 24):
    """The grid build on points in the order a front end delivers them instead of the generator's shuffled order: the
    1M-point config-3 submap room by room, each room's points by bearing around the room's centre (a submap that is a
    sequence of scans), and the 100k-point scan by bearing around the sensor ("the order of the
    points").  Same grids bit for bit (exact integer sums); host call to return."""
    from gtsam_ndt_amd import synth
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3)

    def by_bearing(x, y, cx, cy):
        o = np.argsort(np.arctan2(y - cy, x - cx), kind="stable")
        return x[o].copy(), y[o].copy()

    x, y, L = d["tx"], d["ty"], 50.0
    i, j = np.floor((x + 100.0) / L).clip(0, 3).astype(int), np.floor((y + 100.0) / L).clip(0, 3).astype(int)
    xs, ys = [], []
    for r in range(16):
        sel = (j * 4 + i) == r
        a, b = by_bearing(x[sel], y[sel], (r % 4) * L - 100.0 + 0.5 * L, (r // 4) * L - 100.0 + 0.5 * L)
        xs.append(a); ys.append(b)
    tx, ty = (torch.from_numpy(np.concatenate(v)).to(dev) for v in (xs, ys))
    sx, sy = (torch.from_numpy(v).to(dev) for v in by_bearing(d["sx"], d["sy"], 0.0, 0.0))
    torch.cuda.synchronize()
    with NdtMatcher2D(device=dev_index) as m:
        tb, tu = [], []
        for _ in range(reps):
            t0 = time.perf_counter(); info = m.set_target(tx, ty); tb.append(time.perf_counter() - t0)
        for _ in range(reps):
            t0 = time.perf_counter(); m.add_target_points(sx, sy, pose=d["pose"]); tu.append(time.perf_counter() - t0)
    return {"workload": "config 3 with the submap room by room in bearing order and the scan in bearing order (the shuffled order is grid_build / submap_update)",
            "set_target_ms_per_call": round(1e3 * float(np.median(tb[3:])), 4), "update_ms_per_call": round(1e3 * float(np.median(tu[3:])), 4),
            "n_valid": int(info.n_valid)}


def multi_start_rate(dev_index, tx, ty, sx, sy, init, m: int, steps: int, warmup: int):
    """ndt2d_align_multi_start_dev on the headline pair: m starts around the initial guess carried by one
    launch chain, fixed K iterations each (every start bit-identical to its single-start alignment).
    One call = k_begin + a graph of K + 1 launches + the result fetch (a host sync per call, included)."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    starts = [(init[0] + 0.01 * k, init[1] - 0.01 * k, 0.001 * k) for k in range(m)]
    with NdtMatcher2D(device=dev_index, fixed_iterations=K_GN) as mm:
        mm.set_target(tx, ty)
        for _ in range(max(1, warmup)):
            mm.align_multi_start(sx, sy, starts)
        torch.cuda.synchronize()
        per_call = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            r = mm.align_multi_start(sx, sy, starts)
            per_call.append(time.perf_counter() - t1)
        el = time.perf_counter() - t0
    assert all(q.iterations == K_GN and q.status == 0 for q in r)
    med = float(np.median(per_call))             # a call is ~0.3 ms: one host hiccup (tens of ms) would swamp a mean over few calls
    n = int(sx.numel())
    us = 1e6 * med / (K_GN + 1)
    alg = n * (8 + 24 * m)                      # the points once + one 24 B record per point and start
    return {"starts": m, "iters_per_s_aggregate": round(m * K_GN / med, 1), "ms_per_call": round(1e3 * med, 4),
            "ms_per_call_mean": round(1e3 * el / steps, 4), "timing": f"median of {steps} calls, host call to results on the host",
            "us_per_launch_incl_call_overhead": round(us, 3), "algorithmic_bytes_per_launch": alg,
            "achieved_GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / HBM_PEAK_GBS, 4)}


def multi_scan_rate(dev, dev_index, tx, ty, m: int, steps: int, warmup: int, n_pts: int = 100_000):
    """ndt2d_align_multi_scan_dev: m DIFFERENT 100k-point scans (room (2,1) of the config-3 submap, own sampling
    seeds and poses, generated on the device) against the 1M-point target in one launch chain, fixed K
    iterations each.  Algorithmic bytes per launch = m x N x 32 B (every scan's points and
    records are its own).  One call = k_begin + a graph of K + 1 launches + the result fetch."""
    from gtsam_ndt_amd import synth, synth_dev
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    L, S, tiles = 50.0, 3, 4
    half = 0.5 * tiles * L
    room = synth.room_scene(S + 1000 * (1 * tiles + 2), L, 2 * L - half, 1 * L - half)
    centre = (2 * L - half + 0.5 * L, 1 * L - half + 0.5 * L)
    scans, inits, truth = [], [], []
    for k in range(m):
        pose = (centre[0] + 0.10 - 0.002 * k, centre[1] - 0.08 + 0.002 * k, 0.01 - 0.0002 * k)
        scans.append(synth_dev.sample_scene(room, n_pts, seed=40_000 + k, sigma=synth.SIGMA, pose=pose, device=dev))
        inits.append((centre[0], centre[1], 0.0))
        truth.append(pose)
    with NdtMatcher2D(device=dev_index, fixed_iterations=K_GN) as mm:
        mm.set_target(tx, ty)
        for _ in range(max(1, warmup)):
            mm.align_multi_scan(scans, inits)
        torch.cuda.synchronize()
        per_call = []
        for _ in range(steps):
            t1 = time.perf_counter()
            r = mm.align_multi_scan(scans, inits)
            per_call.append(time.perf_counter() - t1)
    assert all(q.iterations == K_GN and q.status == 0 for q in r)
    err = max(float(np.abs(np.array(q.pose) - np.array(t)).max()) for q, t in zip(r, truth))
    med = float(np.median(per_call))
    us = 1e6 * med / (K_GN + 1)
    alg = m * n_pts * BYTES_PER_POINT_ITER
    return {"scans": m, "points_per_scan": n_pts, "iters_per_s_aggregate": round(m * K_GN / med, 1),
            "ms_per_call": round(1e3 * med, 4), "timing": f"median of {steps} calls, host call to results on the host",
            "us_per_launch_incl_call_overhead": round(us, 3), "algorithmic_bytes_per_launch": alg,
            "achieved_GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / HBM_PEAK_GBS, 4),
            "pose_err_vs_truth_max": err}


def multi_scan_rate_3d(dev, dev_index, target, base_scans, base_poses, m: int, steps: int, warmup: int):
    """ndt3d_align_multi_scan_dev: m different config-5-sized scans (own sensor pose and noise each, ray cast on the
    device) against one cached voxel grid in one launch chain, fixed K iterations each.  Algorithmic bytes per
    iteration = m x N x 52 B."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    scans = [base_scans[k] for k in range(m)]
    inits = [(0.0,) * 6] * m
    with NdtMatcher3D(device=dev_index, fixed_iterations=K_GN) as mm:
        mm.set_target(*target)
        for _ in range(max(1, warmup)):
            mm.align_multi_scan(scans, inits)
        torch.cuda.synchronize()
        per_call = []
        for _ in range(steps):
            t1 = time.perf_counter()
            r = mm.align_multi_scan(scans, inits)
            per_call.append(time.perf_counter() - t1)
    assert all(q.iterations == K_GN and q.status == 0 for q in r)
    errs = [float(np.abs(np.array(q.pose) - np.array(base_poses[k % len(base_poses)])).max()) for k, q in enumerate(r)]
    err = max(errs)
    # the scans that are still > 1 cm from truth after the 30 iterations, and two that are not, against the oracle under
    # the same 30 iterations: kernel == oracle there shows a stray to be the score's basin, not the kernel
    sample = sorted({k for k, e in enumerate(errs) if e >= 1e-2} | {0, m // 2})
    ora = oracle3_fixed30(tuple(c.cpu().numpy() for c in target), [tuple(c.cpu().numpy() for c in scans[k]) for k in sample],
                          [inits[k] for k in sample], cpu_share())
    vs_oracle = max(float(np.abs(np.array(r[k].pose) - np.array(p)).max()) for k, p in zip(sample, ora))
    med = float(np.median(per_call))
    us = 1e6 * med / (K_GN + 1)
    n_pts = int(scans[0][0].numel())
    alg = m * n_pts * 52
    return {"scans": m, "points_per_scan": n_pts, "pose_err_vs_oracle_max": vs_oracle, "scans_checked_vs_oracle": sample, "iters_per_s_aggregate": round(m * K_GN / med, 1),
            "ms_per_call": round(1e3 * med, 4), "timing": f"median of {steps} calls, host call to results on the host",
            "us_per_iteration_incl_call_overhead": round(us, 3), "algorithmic_bytes_per_iteration": alg,
            "achieved_GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / HBM_PEAK_GBS, 4),
            "pose_err_vs_truth_max": err,                  # (single-level 1 m voxels, exactly 30 iterations: a scan that
            "scans_within_1cm_of_truth_after_30_iterations": int(sum(e < 1e-2 for e in errs))}    # starts far is still on its way)


def scan_loop_3d(dev, dev_index, n_scans: int = 40):
    """The 3D front end's loop on the device: 64-beam scans of a sensor moving through the box room,
    each aligned against the submap from an odometry-grade guess (converged mode) and merged into it with the
    estimated pose (ndt3d_reserve_target, ndt3d_align_dev, ndt3d_add_target_points_dev); nothing but the 6-double
    estimate crosses to the host per scan."""
    from gtsam_ndt_amd import synth_dev
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    rng = np.random.default_rng(11)
    poses = [np.zeros(6)]
    for k in range(1, n_scans):                        # 0.3 m and a degree of yaw per scan, small roll / pitch / heave
        step = np.array([0.3 * np.cos(0.02 * k), 0.3 * np.sin(0.02 * k), 0.0, 0.0, 0.0, 0.017])
        p = poses[-1] + step
        p[2] = 0.03 * np.sin(0.3 * k); p[3] = 0.004 * np.sin(0.5 * k); p[4] = 0.004 * np.cos(0.4 * k)
        poses.append(p)
    scans = [synth_dev.lidar_scan3d(700 + k, tuple(p), device=dev, firing_order=True) for k, p in enumerate(poses)]
    torch.cuda.synchronize()
    with NdtMatcher3D(device=dev_index) as m:
        m.reserve_target((-24.0, -24.0, -3.0), (24.0, 24.0, 7.0))
        m.add_target_points(*scans[0], pose=tuple(poses[0]))
        est = [poses[0]]
        errs, iters = [], []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(1, n_scans):
            guess = est[-1] + (poses[k] - poses[k - 1]) + rng.normal(0.0, 1.0, 6) * np.array([0.03, 0.03, 0.01, 0.002, 0.002, 0.005])
            r = m.align(*scans[k], tuple(guess))
            m.add_target_points(*scans[k], pose=r.pose)
            est.append(np.array(r.pose)); iters.append(r.iterations)
            errs.append(float(np.abs(np.array(r.pose) - poses[k])[:3].max()))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        info = m.grid_info()
    return {"workload": f"{n_scans} scans x {int(scans[0][0].numel())} points in firing order, 0.3 m / 1 deg per scan, align (converged) + merge into "
                        "the submap per scan, all on the device",
            "scans_per_s": round((n_scans - 1) / el, 1), "ms_per_scan": round(1e3 * el / (n_scans - 1), 3),
            "iterations_mean": round(float(np.mean(iters)), 1), "position_err_max_m": max(errs), "position_err_last_m": errs[-1],
            "submap_valid_voxels": int(info.n_valid)}


def lidar_batch_rate(dev, dev_index, n_pairs: int = 4096, npts: int = 1000, unique: int = 64):
    """Loop-closure batch of lidar-sized pairs (the 256-thread variant of the batch kernel, two pairs
    per CU): `unique` different synthetic pairs repeated to n_pairs, fixed K iterations each."""
    from gtsam_ndt_amd import dist as nd, synth
    from gtsam_ndt_amd.matcher import NdtBatch2D
    pairs = [synth.make_pair(4, pair_index=20000 + k, n_tgt=npts, n_src=npts) for k in range(unique)]
    pairs = (pairs * ((n_pairs + unique - 1) // unique))[:n_pairs]
    t = {k: torch.from_numpy(v).to(dev) for k, v in nd.pack_pairs(pairs).items()}
    with NdtBatch2D(device=dev_index, fixed_iterations=K_GN) as b:
        side = torch.cuda.ExternalStream(b.stream)
        best = None
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            out = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"])
            e1.record(side)
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        rows = NdtBatch2D.decode(out)
    ok = sum(r.status == 0 and r.iterations == K_GN for r in rows)
    return {"workload": f"{n_pairs} pairs x {npts}-point scans ({unique} distinct), fixed {K_GN} iterations per pair",
            "ms_per_batch": round(best, 4), "pairs_per_s": round(n_pairs / (best * 1e-3), 1),
            "pair_iterations_per_s": round(n_pairs * K_GN / (best * 1e-3), 1), "pairs_ok": ok}


def overlap_batch_rate(dev, dev_index, n_pairs: int = 256, npts: int = 100_000):
    """The loop-closure batch with Biber's four overlapping grids (overlap_grids = 4: every pair on the global-table
    variant, the four grids taking turns in LDS) on config-4 pairs generated in HBM; one pair is
    checked against the single-pair path with the same option."""
    from gtsam_ndt_amd import synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    t = synth_dev.config4_batch(0, n_pairs, npts, npts, device=dev)
    with NdtBatch2D(device=dev_index, fixed_iterations=K_GN, overlap_grids=4) as b:
        side = torch.cuda.ExternalStream(b.stream)
        best = None
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            out = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"])
            e1.record(side)
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        rows = NdtBatch2D.decode(out)
    with NdtMatcher2D(device=dev_index, fixed_iterations=K_GN, overlap_grids=4) as m:
        m.set_target(t["tx"][:npts], t["ty"][:npts])
        r0 = m.align(t["sx"][:npts], t["sy"][:npts], tuple(t["init"][0].tolist()))
    err = max(abs(u - v) for u, v in zip(rows[0].pose, r0.pose))
    return {"workload": f"{n_pairs} config-4 pairs ({npts}/{npts} points), overlap_grids = 4, fixed {K_GN} iterations per pair",
            "ms_per_batch": round(best, 4), "pair_iterations_per_s": round(n_pairs * K_GN / (best * 1e-3), 1),
            "pairs_ok": sum(r.status == 0 and r.iterations == K_GN for r in rows), "pose_err_vs_single_pair": float(f"{err:.3g}")}


def load_traffic():
    """HBM bytes per k_iterate launch from the committed PMC profile (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


def cpu_share() -> int:
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box hands a one-GPU job 16 cores of a much larger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(d, seconds: float):
    """The C restatement of the oracle (kind "port"), built here with -O3 -march=native and timed
    on this host's cores on a bounded sample of the same workload: fixed-K alignments of the
    config-3 pair until `seconds`, single-threaded and on every core this job may use."""
    from gtsam_ndt_amd import build
    from oracle import cport, ndt2d
    lib_path = build.build_oracle(native=True)
    prm = ndt2d.NdtParams(fixed_iterations=K_GN)
    t0 = time.perf_counter()
    g = cport.CGrid(d["tx"], d["ty"], prm, lib_path=lib_path)
    grid_s = time.perf_counter() - t0
    best = None
    runs = []
    share = cpu_share()
    nthr = max(1, min(int(cport.load(lib_path).orc_max_threads()), share))
    counts = sorted({1, nthr})
    for threads in counts:
        it = 0
        n_al = 0
        g.align(d["sx"], d["sy"], d["init"], threads=threads)          # warm the thread pool and the caches
        t0 = time.perf_counter()
        while True:
            r = g.align(d["sx"], d["sy"], d["init"], threads=threads)
            it += r["iterations"]
            n_al += 1
            el = time.perf_counter() - t0
            if el >= seconds / len(counts):
                break
        rate = it / el
        runs.append({"cores": threads, "value": round(rate, 1)})
        if best is None or rate > best["value"]:
            best = {"value": round(rate, 1), "unit": "iters/s", "cores": threads, "kind": "port",
                    "sample": f"{n_al} fixed-K={K_GN} alignments of the same config-3 pair "
                              f"({el:.1f} s of CPU work), oracle/ndt_oracle.c (gcc -O3 -march=native -fopenmp, built on this "
                              f"host) with {threads} thread(s) of the {share} cores this job may use; "
                              f"1M-point grid build {grid_s * 1e3:.0f} ms excluded, as on the GPU"}
    g.close()
    best["runs"] = runs
    return best


def cross_check_batch_rows(dev_index, t, rows, ppr, every: int = 64):
    """Parity inside the bench: every `every`-th pair of this rank's shard is aligned again through the
    single-pair path (k_iterate, its own grid build in global memory) with the same fixed K, and the
    batch kernel's row is compared with it.  Outside the timed region."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    worst = np.zeros(3)
    checked = 0
    toff, soff = t["toff"].cpu().numpy(), t["soff"].cpu().numpy()
    init = t["init"].cpu().numpy()
    with NdtMatcher2D(device=dev_index, fixed_iterations=K_GN) as m:
        for k in range(0, ppr, every):
            m.set_target(t["tx"][toff[k]:toff[k + 1]], t["ty"][toff[k]:toff[k + 1]])
            r = m.align(t["sx"][soff[k]:soff[k + 1]], t["sy"][soff[k]:soff[k + 1]], init[k])
            assert r.status == rows[k].status == 0 and r.iterations == rows[k].iterations == K_GN
            worst = np.maximum(worst, np.abs(np.array(r.pose) - np.array(rows[k].pose)))
            checked += 1
    return {"dx_m": float(worst[0]), "dy_m": float(worst[1]), "dtheta_rad": float(worst[2]), "pairs_checked": checked,
            "note": f"every {every}th pair re-aligned through the single-pair path (k_iterate), fixed {K_GN} iterations: "
                    "max |pose difference| between the two kernels"}


def run_batch(a, dev, dev_index, rank, world, dist, barrier, ppr=None, steps=None, converged=False):
    """BASELINE config 4, this rank's shard: `ppr` candidate pairs x batch_points, generated in HBM
    by the device twin of synth.py (bit-identical to synth.make_pair(4, k)); a step aligns every
    pair (LDS grid build + 30 GN iterations each) and gathers the per-pair results of all ranks
    (RCCL all_gather; the path's only collective)."""
    from gtsam_ndt_amd import dist as nd, synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch2D
    ppr = ppr or a.pairs_per_rank
    steps = steps or a.steps
    npts = a.batch_points
    total = ppr * world
    # fixed-K mode: every pair costs the same, contiguous shards.  Converged mode: pairs need different iteration
    # counts, so the shards are strided (pair k -> rank k mod world) and the gather undoes the permutation.
    strided = bool(converged and world > 1)
    mine = nd.shard_range(total, rank, world, strided=strided)
    t0 = time.perf_counter()
    t = synth_dev.config4_batch(mine.start, len(mine), npts, npts, device=dev, indices=list(mine) if strided else None)
    torch.cuda.synchronize()
    gen_ms = 1e3 * (time.perf_counter() - t0)
    truth = t["pose"].cpu().numpy()
    b = NdtBatch2D(device=dev_index, fixed_iterations=0 if converged else K_GN)
    host_ms = None
    if dist is None and a.host_path and ppr <= 512:
        # the same batch through the host-pointer entry point (pageable host arrays -> upload ->
        # kernel -> results back): the PCIe-inclusive figure DESIGN.md section 7 quotes; never `value`
        from gtsam_ndt_amd import _lib as L
        from gtsam_ndt_amd.matcher import RESULT_DOUBLES
        h = {k: np.ascontiguousarray(v.cpu().numpy()) for k, v in t.items()}
        outh = np.zeros(ppr * RESULT_DOUBLES, dtype=np.float64)
        lat = []
        for _ in range(3):
            t0 = time.perf_counter()
            L.check(b._lib.ndt2d_batch_align(b._h, h["tx"].ctypes.data, h["ty"].ctypes.data, h["toff"].ctypes.data,
                                             h["sx"].ctypes.data, h["sy"].ctypes.data, h["soff"].ctypes.data,
                                             h["init"].ctypes.data, ppr, outh.ctypes.data), "ndt2d_batch_align")
            lat.append(time.perf_counter() - t0)
        host_ms = 1e3 * min(lat[1:])
        del h
    # one explicit (non-default) stream carries the kernel, the HIP events and - through
    # torch.distributed's stream ordering - the RCCL all_gather that consumes the results
    side = torch.cuda.Stream(device=dev)
    res = torch.empty((ppr, 18), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        cur = side.cuda_stream
        assert cur != 0

        def step():
            b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"], out=res, stream=cur)
            return nd.gather_results(res, total, strided=strided) if dist is not None else res

        for _ in range(max(1, a.warmup)):
            allr = step()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for e0, e1 in ev:
            e0.record(side)
            b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"], out=res, stream=cur)
            e1.record(side)
            allr = nd.gather_results(res, total, strided=strided) if dist is not None else res
        barrier()
        elapsed = time.perf_counter() - t0
        kern_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev)
    if dist is not None:
        elapsed = nd.max_over_ranks(elapsed, device=dev)
    rows = NdtBatch2D.decode(allr)
    local = [rows[k] for k in mine]
    err = np.abs(np.array([r.pose for r in local]) - truth)      # vs the generating pose (sampling noise)
    if converged:
        return converged_batch_line(a, rows, local, err, elapsed, kern_ms, steps, world, ppr, total, npts, strided, dist, dev, nd)
    assert len(rows) == total and all(r.iterations == K_GN and r.status == 0 for r in rows)
    cross = cross_check_batch_rows(dev_index, t, local, ppr) if not a.headline_only else None
    if dist is not None and cross is not None:
        for k in ("dx_m", "dy_m", "dtheta_rad"):
            cross[k] = nd.max_over_ranks(cross[k], device=dev)
    iters = total * K_GN * steps
    launch_ms = kern_ms / steps
    # Algorithmic HBM bytes of one pair for THIS kernel: the target once
    # (8 B/pt) + the source once per iteration (8 B/pt; 0.8 MB per pair does not fit on chip)
    # + one result row.  SURVEY.md section 8d's figure of record, 32 B per point-iteration,
    # also counts the 24 B cell-record gather, which this kernel serves from LDS - it is
    # reported beside it, not as the roofline numerator.
    alg_pair = 8 * npts + K_GN * 8 * npts + 144
    streamed_pair = 3 * 8 * npts + K_GN * 8 * npts + 144          # what the kernel reads: 3 target passes
    survey_pair = 8 * npts + 24 * 10816 + K_GN * BYTES_PER_POINT_ITER * npts
    achieved = ppr * alg_pair / (launch_ms * 1e-3) / 1e9
    traffic = load_traffic() or {}
    out = {
        "metric": METRIC, "value": round(iters / elapsed, 1), "unit": "iters/s", "n_gpus": world,
        "steps": steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config4: loop-closure batch, {ppr} candidate pairs x {npts} pts per GPU "
                               f"({total} pairs total), 0.5 m cells, grid build + fixed 30 GN iterations per pair",
                   "pairs_per_gpu": ppr, "pairs_total": total, "n_target": npts, "n_source": npts,
                   "cell_size": 0.5, "gn_iterations_per_pair": K_GN,
                   "collective": ("all_gather of 144 B/pair inside every timed step ("
                                  + ("gloo on host copies: one-GPU rehearsal" if a.rehearse_on_one_gpu else "RCCL") + ")")
                   if dist is not None else "none (1 GPU)",
                   "generator": f"device twin of synth.make_pair(4, k), k = {mine.start}..{mine.stop - 1} on this rank "
                                f"({gen_ms:.0f} ms incl. first-use overheads; outside the timed region)"},
        "pairs_per_s": round(total * steps / elapsed, 1),
        "scaling_note": "weak scaling of the loop-closure batch: compare with the N=1 line's batch.value",
        "roofline": {"bound": "hbm", "kernel": "k_batch<GN>", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic.get("batch_bytes_per_launch") if ppr == 512 else None,
                     "traffic_source": traffic.get("source", "profiles/ (rocprofv3 --pmc, separate run of this command)"),
                     "algorithmic_bytes_per_launch": ppr * alg_pair, "avg_launch_us": round(1e3 * launch_ms, 1),
                     "streamed_bytes_per_launch": ppr * streamed_pair,
                     "survey_8d_bytes_per_launch": ppr * survey_pair,
                     "survey_8d_GBps": round(ppr * survey_pair / (launch_ms * 1e-3) / 1e9, 1),
                     "timing": "HIP events on the launch stream around each k_batch launch (rank 0)",
                     "note": "numerator = 8 B/pt target once + 8 B/pt source per iteration; SURVEY.md 8d's "
                             "32 B per point-iteration also counts the 24 B cell record, LDS-served here "
                             "(survey_8d_* fields)"},
        "pose_err_vs_truth_max": {"dx_m": float(err[:, 0].max()), "dy_m": float(err[:, 1].max()),
                                  "dtheta_rad": float(err[:, 2].max()),
                                  "note": "fixed 30 iterations vs the generating pose (sampling noise included)"},
        "pose_err_vs_single_pair_max": cross,
    }
    if host_ms is not None:
        nbytes = ppr * npts * 16
        out["host_path"] = {"ms_per_step_incl_pcie": round(host_ms, 2),
                            "iters_per_s_incl_pcie": round(ppr * K_GN / (host_ms * 1e-3), 1),
                            "uploaded_MB": round(nbytes / 1e6, 1),
                            "note": "ndt2d_batch_align with pageable host arrays: upload + kernel + results; "
                                    "reported beside value, never as value"}
    b.close()
    del t, res
    torch.cuda.empty_cache()
    return out


def converged_batch_line(a, rows, local, err, elapsed, kern_ms, steps, world, ppr, total, npts, strided, dist, dev, nd):
    """The `batch_converged` leg: every pair runs until its own convergence test passes.  value counts the
    Gauss-Newton iterations actually made.  With N > 1 the shards are strided and `shard_balance` shows the
    per-rank iteration totals of the strided split next to what contiguous shards of the same rows would carry."""
    its = np.array([r.iterations for r in rows], dtype=np.int64)
    ok = int(sum(r.status == 0 for r in rows))
    by_rank_strided = [int(its[list(nd.shard_range(total, r, world, strided=True))].sum()) for r in range(world)]
    by_rank_contig = [int(its[list(nd.shard_range(total, r, world))].sum()) for r in range(world)]
    return {"workload": f"config4 in converged mode: {ppr} candidate pairs x {npts} pts per GPU ({total} total), each pair until "
                        "its own convergence test passes (max 100 iterations)",
            "value": round(float(its.sum()) * steps / elapsed, 1), "unit": "iters/s (iterations actually made)",
            "pairs_per_s": round(total * steps / elapsed, 1), "ms_per_step": round(1e3 * elapsed / steps, 4), "steps": steps,
            "n_gpus": world, "sharding": "strided (pair k -> rank k mod N)" if strided else "single GPU",
            "pairs_converged": ok, "iterations_min_mean_max": [int(its.min()), round(float(its.mean()), 1), int(its.max())],
            "shard_balance": {"iterations_per_rank_strided": by_rank_strided, "iterations_per_rank_contiguous": by_rank_contig,
                              "max_over_mean_strided": round(max(by_rank_strided) / max(1.0, float(np.mean(by_rank_strided))), 4),
                              "max_over_mean_contiguous": round(max(by_rank_contig) / max(1.0, float(np.mean(by_rank_contig))), 4)},
            "avg_launch_ms_rank0": round(kern_ms / steps, 3),
            "pose_err_vs_truth_max": {"dx_m": float(err[:, 0].max()), "dy_m": float(err[:, 1].max()), "dtheta_rad": float(err[:, 2].max())}}


def oracle3_fixed30(target_host, scans_host, inits, threads):
    """Checker leg (outside every timed region): the C twin of oracle/ndt3d.py (built here with -O3 -march=native)
    aligns each of `scans_host` against the voxel grid of `target_host` - or of its own target when target_host is a
    list - with the same fixed K_GN iterations the kernels ran.  Returns the oracle's poses."""
    from gtsam_ndt_amd import build
    from oracle import cport, ndt3d
    lib_path = build.build_oracle(native=True)
    prm = ndt3d.Ndt3Params(fixed_iterations=K_GN)
    poses = []
    shared = None if isinstance(target_host, list) else cport.CGrid3(*target_host, prm, lib_path=lib_path)
    for k, sc in enumerate(scans_host):
        g = shared or cport.CGrid3(*target_host[k], prm, lib_path=lib_path)
        r = g.align(*sc, inits[k], threads=threads)
        assert r["iterations"] == K_GN
        poses.append(r["pose"])
        if shared is None:
            g.close()
    if shared is not None:
        shared.close()
    return poses


def cpu_baseline_3d(d, seconds: float):
    """SURVEY 8d 'report for every config': the C twin of the 3D oracle (kind "port") timed on this host's cores on a
    bounded sample of config 5 - fixed-K alignments of the same 131 072-point pair until `seconds`."""
    from gtsam_ndt_amd import build
    from oracle import cport, ndt3d
    lib_path = build.build_oracle(native=True)
    prm = ndt3d.Ndt3Params(fixed_iterations=K_GN)
    t0 = time.perf_counter()
    g = cport.CGrid3(d["tx"], d["ty"], d["tz"], prm, lib_path=lib_path)
    grid_s = time.perf_counter() - t0
    share = cpu_share()
    nthr = max(1, min(int(cport.load(lib_path).orc_max_threads()), share))
    runs, best = [], None
    for threads in sorted({1, nthr}):
        g.align(d["sx"], d["sy"], d["sz"], d["init"], threads=threads)
        it = n_al = 0
        t0 = time.perf_counter()
        while True:
            r = g.align(d["sx"], d["sy"], d["sz"], d["init"], threads=threads)
            it += r["iterations"]; n_al += 1
            el = time.perf_counter() - t0
            if el >= seconds / 2:
                break
        rate = it / el
        runs.append({"cores": threads, "value": round(rate, 1)})
        if best is None or rate > best["value"]:
            best = {"value": round(rate, 1), "unit": "iters/s", "cores": threads, "kind": "port",
                    "sample": f"{n_al} fixed-K={K_GN} alignments of the same config-5 pair ({el:.1f} s of CPU work), "
                              f"oracle/ndt_oracle.c orc3d_* (gcc -O3 -march=native -fopenmp, built on this host) with {threads} thread(s) "
                              f"of the {share} cores this job may use; voxel grid build {grid_s * 1e3:.0f} ms excluded, as on the GPU"}
    g.close()
    best["runs"] = runs
    return best


def run_3d(a, dev, dev_index):
    """BASELINE config 5: 3D SE(3), 64 x 2048 beams, fixed 30 Gauss-Newton iterations per step, timed as the
    2D headline is: alignments enqueued back to back (ndt3d_align_dev_async), HIP events on the handle's
    stream over the timed region, one fetch at the end."""
    from gtsam_ndt_amd import synth3d
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d = synth3d.make_pair3d()
    s = [torch.from_numpy(d[k]).to(dev) for k in ("sx", "sy", "sz")]
    with NdtMatcher3D(device=dev_index, fixed_iterations=K_GN) as m:
        t = [torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "tz")]
        torch.cuda.synchronize()
        tg = []
        for _ in range(4):
            t0 = time.perf_counter(); m.set_target(*t); tg.append(time.perf_counter() - t0)
        grid_ms = 1e3 * float(np.median(tg[1:]))
        for _ in range(max(1, a.warmup)):
            m.align_async(*s, d["init"], producer_complete=True)
        m.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_ms = hip_events_ms(m.stream, lambda: [m.align_async(*s, d["init"], producer_complete=True) for _ in range(a.steps)])
        r = m.finish()
        el = time.perf_counter() - t0
    assert r.iterations == K_GN and r.status == 0
    n = int(s[0].numel())
    per_launch_us = 1e3 * ev_ms / (a.steps * (K_GN + 1))
    alg = n * 52                                            # 12 B point + 40 B record
    multi = None
    if not a.headline_only:
        from gtsam_ndt_amd import synth_dev
        rng = np.random.default_rng(5)
        poses = [tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-0.5, 0.5, 6)) for _ in range(64)]
        base = [synth_dev.lidar_scan3d(300 + k, p, device=dev) for k, p in enumerate(poses)]     # 64 distinct scans
        multi = {"note": "the same cached voxel grid, m different config-5-sized scans per launch chain (ndt3d_align_multi_scan_dev; "
                         "own sensor pose and noise per scan, ray cast on the device); beside the single-scan figure, never instead of it; "
                         "bytes = m x N x 52 B",
                 "runs": [multi_scan_rate_3d(dev, dev_index, t, base, poses, mm_, max(5, a.steps // 2), a.warmup) for mm_ in (8, 64)]}
    cpu3 = None if (a.no_cpu_baseline or a.headline_only) else cpu_baseline_3d(d, min(a.cpu_seconds, 8.0))
    return {"workload": "config5: 3D NDT SE(3), 131072-pt synthetic 64-beam scans, 1.0 m cells, fixed 30 GN iterations",
            "cpu_baseline": cpu3,
            "multi_scan": multi,
            "value": round(a.steps * K_GN / el, 1), "unit": "iters/s", "ms_per_step": round(1e3 * el / a.steps, 4),
            "grid_build_ms": round(grid_ms, 3), "iterations": r.iterations,
            "pose_after_30": list(r.pose), "true_pose": list(d["pose"]),
            "roofline": {"bound": "hbm", "kernel": "k_iterate3", "algorithmic_bytes_per_launch": alg,
                         "traffic": (load_traffic() or {}).get("bytes_per_launch_3d"),
                         "avg_launch_us": round(per_launch_us, 3),
                         "timing": "HIP events on the handle's stream over the timed region / launches (kernel + launch boundary)",
                         "achieved": round(alg / (per_launch_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (per_launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}


def run_batch_3d(a, dev, dev_index, n_pairs=256, check=True):
    """3D loop-closure batch (ndt3d_batch_align_dev): n_pairs config-5-sized pairs (131072 points each) per step,
    fixed 30 iterations per pair.  Every pair is its own scene (clutter boxes from scene seed 5 + k), its own noise
    and its own relative pose, ray cast on the device (synth_dev.lidar_scan3d: the device twin of synth3d.lidar_scan)."""
    from gtsam_ndt_amd import synth3d, synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    rng = np.random.default_rng(5)
    poses = [tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-0.5, 0.5, 6)) for _ in range(n_pairs)]
    n_elev, n_azim = 64, 2048
    npts = n_elev * n_azim
    t = [torch.empty(n_pairs * npts, dtype=torch.float32, device=dev) for _ in range(3)]
    s = [torch.empty(n_pairs * npts, dtype=torch.float32, device=dev) for _ in range(3)]
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * npts
    init = torch.zeros((n_pairs, 6), dtype=torch.float64, device=dev)
    steps = max(3, min(a.steps, 10))

    def run(firing_order: bool):
        """Generate the pairs in the given point order (the same points either way) and time the batch on them."""
        t_gen = time.perf_counter()
        for k, p in enumerate(poses):
            sl = slice(k * npts, (k + 1) * npts)
            synth_dev.lidar_scan3d(1000 + 2 * k, (0.0,) * 6, n_elev, n_azim, 0.02, scene_seed=5 + k, out=tuple(c[sl] for c in t),
                                   firing_order=firing_order)
            synth_dev.lidar_scan3d(1001 + 2 * k, p, n_elev, n_azim, 0.02, scene_seed=5 + k, out=tuple(c[sl] for c in s),
                                   firing_order=firing_order)
        torch.cuda.synchronize()
        gen = 1e3 * (time.perf_counter() - t_gen)
        with NdtBatch3D(device=dev_index, fixed_iterations=K_GN) as b:
            out = None
            for _ in range(max(1, min(a.warmup, 2))):
                out = b.align_dev(t, off, s, off, init, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev = hip_events_ms(b.stream, lambda: [b.align_dev(t, off, s, off, init, out=out, stream=b.stream) for _ in range(steps)])
            torch.cuda.synchronize()
            return time.perf_counter() - t0, ev, b.decode(out), gen

    # the figure is quoted on scans in firing order (all 64 beams of one bearing, then the next bearing: what a spinning
    # lidar's driver delivers); the same points ring by ring are timed beside it (the build's LDS atomics collide more)
    el_ring, _, res_ring, _ = run(False)
    assert all(r.status == 0 and r.iterations == K_GN for r in res_ring)
    el, ev_ms, res, gen_ms = run(True)
    assert all(r.status == 0 and r.iterations == K_GN for r in res)
    # sampled cross-check against the single-pair path (k_iterate3): every 64th pair
    cross, checked = None, 0
    with NdtMatcher3D(device=dev_index, fixed_iterations=K_GN) as m:
        for k in range(0, n_pairs if check else 0, 64):
            sl = slice(k * npts, (k + 1) * npts)
            m.set_target(*(c[sl].cpu().numpy() for c in t))
            r1 = m.align(*(c[sl].contiguous() for c in s), (0.0,) * 6)
            cross = max(cross or 0.0, float(np.abs(np.array(r1.pose) - np.array(res[k].pose)).max()))
            checked += 1
    errs = np.array([float(np.abs(np.array(res[k].pose) - np.array(poses[k])).max()) for k in range(n_pairs)])
    err = float(errs.max())
    vs_oracle, sample = None, []
    if check:
        # the pairs that are still > 1 cm from truth after the 30 iterations (and every 64th pair) against the float64
        # oracle under the same 30 iterations (tests/test_gpu_config5_fullsize.py asserts the same at 1e-4)
        sample = sorted({int(k) for k in np.nonzero(errs >= 1e-2)[0]} | set(range(0, n_pairs, 64)))
        host = lambda arrs, k: tuple(c[k * npts:(k + 1) * npts].cpu().numpy() for c in arrs)
        ora = oracle3_fixed30([host(t, k) for k in sample], [host(s, k) for k in sample], [(0.0,) * 6] * len(sample), cpu_share())
        vs_oracle = max(float(np.abs(np.array(res[k].pose) - np.array(p)).max()) for k, p in zip(sample, ora))
    launch_ms = ev_ms / steps
    alg = n_pairs * npts * 12 * (1 + K_GN)                  # target once + source once per iteration, 12 B per point
    return {"workload": f"{n_pairs} distinct 3D scan pairs of config-5 size ({npts} + {npts} points in firing order, 1.0 m voxels; own "
                        f"scene, noise and pose per pair, generated on the device in {gen_ms:.0f} ms), fixed 30 GN iterations per pair, one GPU",
            "value": round(n_pairs * K_GN * steps / el, 1), "unit": "pair-iterations/s",
            "pairs_per_s": round(n_pairs * steps / el, 1), "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
            "pose_diff_vs_single_pair_max": cross, "pairs_checked_vs_single_pair": checked, "pose_err_vs_truth_max": err,
            "pose_err_vs_oracle_max": vs_oracle, "pairs_checked_vs_oracle": sample,
            "pairs_within_1cm_of_truth_after_30_iterations": int((errs < 0.01).sum()),
            "point_order": "firing order (index = bearing * 64 + beam)",
            "same_points_ring_by_ring": {"value": round(n_pairs * K_GN * steps / el_ring, 1), "ms_per_step": round(1e3 * el_ring / steps, 3),
                                         "pose_diff_vs_firing_order_max": float(max(np.abs(np.array(x.pose) - np.array(y.pose)).max()
                                                                                    for x, y in zip(res, res_ring)))},
            "roofline": {"bound": "hbm", "kernel": "k_batch3<GN>", "algorithmic_bytes_per_launch": alg,
                         "traffic": (load_traffic() or {}).get("batch3_bytes_per_launch") if n_pairs == 256 else None,
                         "bytes_rule": "12 B x (target points + 30 x source points) per pair: the 40-byte voxel records "
                                       "are served from LDS (SURVEY 8d's 52 B per point counts them as HBM traffic)",
                         "avg_launch_us": round(1e3 * launch_ms, 1),
                         "timing": "HIP events on the context's stream over the timed region / launches",
                         "achieved": round(alg / (launch_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "bound by VALU issue, not by memory: 127 vector instructions per point and iteration "
                                 "(DESIGN.md section 5.5)"}}


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def self_launch(a) -> int:
    """--gpus N > 1 without a launcher: start N ranks of this script as a CHILD `python -m torch.distributed.run`
    (one process per GPU, rendezvous on 127.0.0.1 at a free port), relay rank 0's JSON line to stdout and everything
    else to stderr, and return the child's exit code.  Called before anything in this process touches the GPU, and
    the child is a fresh process - nothing that has initialised HIP is ever re-executed."""
    import subprocess
    if not a.rehearse_on_one_gpu:
        have = torch.cuda.device_count()          # counting devices does not initialise the GPU on this image
        if have < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {have} device(s) visible "
                  "(a one-GPU rehearsal of the N > 1 flow needs --rehearse-on-one-gpu)", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it)
    line = None
    with subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=ROOT) as p:
        for out_line in p.stdout:
            if out_line.startswith("{") and '"metric"' in out_line:
                line = out_line.strip()
            else:
                sys.stderr.write(out_line)
        rc = p.wait()
    if rc != 0:
        print(f"bench.py: the {a.gpus}-rank child run failed with exit code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the child run printed no JSON line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.rehearse_on_one_gpu:
            local_rank = 0
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if (world > 1 and not a.rehearse_on_one_gpu) else 0
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from gtsam_ndt_amd import _lib, synth
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    if _lib.load().ndt_device_count() < 1:
        raise RuntimeError("bench.py needs a gfx950 device: the NDT matcher has no CPU fallback")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    if dist is None:
        # ------------------------------------------------------------- config 3, single pair
        d = synth.make_pair(3)
        tx, ty, sx, sy = (torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "sx", "sy"))
        torch.cuda.synchronize()
        m = NdtMatcher2D(device=dev_index, fixed_iterations=K_GN)
        grid_build, submap_update = build_legs(m, tx, ty, sx, sy, d["pose"])
        grid_ms = grid_build["ms_per_call"]
        m.set_target(tx, ty)                   # (the timed alignments run against the plain 1M-point submap again)
        n_src = int(sx.numel())

        def step():
            m.align_async(sx, sy, d["init"], producer_complete=True)     # inputs resident and complete since the synchronise above

        for _ in range(a.warmup):
            step()
        m.finish()
        barrier()
        t0 = time.perf_counter()
        ev_ms = hip_events_ms(m.stream, lambda: [step() for _ in range(a.steps)])
        r = m.finish()
        barrier()
        elapsed = time.perf_counter() - t0
        assert r.iterations == K_GN
        iters = a.steps * K_GN
        launches = a.steps * (K_GN + 1)
        value = iters / elapsed
        launch_us = 1e3 * ev_ms / launches
        alg_bytes = n_src * BYTES_PER_POINT_ITER
        achieved = alg_bytes / (launch_us * 1e-6) / 1e9
        traffic = load_traffic()
        roofline = {"bound": "hbm", "kernel": "k_iterate<GN>", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": (traffic or {}).get("bytes_per_launch"),
                    "traffic_source": "not measured in this run: read from the committed PMC passes of the same command, "
                                      + (traffic or {}).get("source", "profiles/pmc_traffic.json"),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_launch_us": round(launch_us, 3),
                    "timing": "HIP events on the handle's stream over the timed region / launches "
                              "(kernel + launch boundary)"}
        # converged-pose error vs the CPU oracle (outside the timed region)
        from oracle import ndt2d as oracle
        rc = ref = perr = conv_ms = relaxed = None
        if not a.headline_only:
            mc = NdtMatcher2D(device=dev_index)
            mc.set_target(tx, ty)
            rc = mc.align(sx, sy, d["init"])
            # what a caller sees per scan in converged mode: host call -> result on the host
            lat = []
            for _ in range(0 if a.no_latency else 20):
                t1 = time.perf_counter(); rl = mc.align(sx, sy, d["init"]); lat.append(time.perf_counter() - t1)
                assert rl.pose == rc.pose
            conv_ms = 1e3 * float(np.median(lat)) if lat else None
            mc.close()
            if not a.no_latency:
                # the same call with over-relaxed steps (params.step_scale = 3)
                mr = NdtMatcher2D(device=dev_index, step_scale=3.0)
                mr.set_target(tx, ty)
                lat = []
                for _ in range(20):
                    t1 = time.perf_counter(); rr = mr.align(sx, sy, d["init"]); lat.append(time.perf_counter() - t1)
                mr.close()
                relaxed = {"step_scale": 3.0, "ms_per_call": round(1e3 * float(np.median(lat)), 4), "iterations": rr.iterations,
                           "status": rr.status,
                           "pose_diff_vs_plain": [abs(u - v) for u, v in zip(rr.pose, rc.pose)]}
            prm = oracle.NdtParams()
            ref = oracle.align(oracle.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
            perr = np.abs(np.array(rc.pose) - np.array(ref["pose"]))
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "iters/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config3: 2D NDT, 1M-pt submap target vs 100k-pt scan, single pair, "
                                   "0.5 m cells, fixed 30 GN iterations per step",
                       "n_target": int(tx.numel()), "n_source": n_src, "cell_size": 0.5,
                       "gn_iterations_per_step": K_GN, "hessian": "gauss-newton"},
            "roofline": roofline,
            "grid_build_ms": round(grid_ms, 4),
            "grid_build": grid_build,
            "submap_update": submap_update,
            "submap_update_ms": submap_update["ms_per_call"],
            "converged_align": None if rc is None else {"ms_per_call": None if conv_ms is None else round(conv_ms, 4), "iterations": rc.iterations,
                                "note": "ndt2d_align_dev in converged mode, host call to result in host memory "
                                        "(8-launch chunks, progress and done flag written to pinned host memory); median of 20",
                                "relaxed": relaxed},
            "scaling_note": "N > 1 lines carry this workload as one independent replica per GPU in `value` (a single "
                            "alignment does not shard) and the sharded loop-closure batch of config 4 in `batch`: "
                            "read value against value and batch.value against batch.value.",
            "pose_err_vs_cpu_ref": None if perr is None else {
                "dx_m": float(perr[0]), "dy_m": float(perr[1]), "dtheta_rad": float(perr[2]),
                "gpu_iterations": rc.iterations, "cpu_iterations": ref["iterations"],
                "cpu_ref": "oracle/ndt2d.py float64 (reference implementation unavailable)"},
        }
        m.close()
        if not a.headline_only:
            out["multi_start"] = {
                "note": "the same 1M-point target and 100k-point scan, m starts per launch chain (ndt2d_align_multi_start_dev); "
                        "reported beside the single-start headline, never instead of it; bytes = N x (8 + 24 m)",
                "runs": [multi_start_rate(dev_index, tx, ty, sx, sy, d["init"], mm_, max(5, a.steps // 2), a.warmup)
                         for mm_ in [int(v) for v in a.multi_starts.split(",") if v]]}
        if not a.headline_only:
            out["multi_scan"] = {
                "note": "the same 1M-point target, m different 100k-point scans per launch chain (ndt2d_align_multi_scan_dev); "
                        "beside the single-scan headline, never instead of it; bytes = m x N x 32 B",
                "runs": [multi_scan_rate(dev, dev_index, tx, ty, mm_, max(5, a.steps // 2), a.warmup) for mm_ in (8, 64)]}
        if a.host_path:
            mh = NdtMatcher2D(device=dev_index, fixed_iterations=K_GN)
            lt, la = [], []
            for _ in range(5):
                t1 = time.perf_counter(); mh.set_target(d["tx"], d["ty"]); lt.append(time.perf_counter() - t1)
            for _ in range(10):
                t1 = time.perf_counter(); rh = mh.align(d["sx"], d["sy"], d["init"]); la.append(time.perf_counter() - t1)
            mh.close()
            assert rh.iterations == K_GN
            out["host_path"] = {"set_target_ms_incl_pcie": round(1e3 * float(np.median(lt[1:])), 4),
                                "align_ms_incl_pcie": round(1e3 * float(np.median(la[1:])), 4),
                                "iters_per_s_incl_pcie": round(K_GN / float(np.median(la[1:])), 1),
                                "note": "ndt2d_set_target / ndt2d_align with pageable host arrays (8 MB / 0.8 MB "
                                        "uploads included); reported beside value, never as value"}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, a.cpu_seconds)
        if not a.no_batch:
            out["batch"] = run_batch(a, dev, dev_index, 0, 1, None, barrier)
            if a.converged_batch:
                out["batch_converged"] = run_batch(a, dev, dev_index, 0, 1, None, barrier, steps=max(3, min(a.steps, 10)), converged=True)
            if not a.no_batch_4096:
                # all 4096 candidates of BASELINE config 4 on this one GPU (6.55 GB resident): the
                # anchor for reading the N > 1 lines (512 pairs per GPU) as strong scaling as well
                out["batch_4096"] = run_batch(a, dev, dev_index, 0, 1, None, barrier, ppr=4096, steps=max(3, min(a.steps, 10)))
        if not a.no_3d:
            out["3d"] = run_3d(a, dev, dev_index)
            out["batch_3d"] = run_batch_3d(a, dev, dev_index, check=not a.headline_only)   # profile mode: timed launches only
        # the other single-pair configs of BASELINE.json beside the headline (parity-test cases; cheap to time)
        if not a.headline_only:
            out["configs_1_2"] = [single_pair_rate(dev, dev_index, c, a.steps, a.warmup) for c in (1, 2)]
        if a.all_configs:
            out["batch_lidar_sized"] = lidar_batch_rate(dev, dev_index)
            out["batch_overlap_grids"] = overlap_batch_rate(dev, dev_index)
            out["grid_build_ordered"] = ordered_build_leg(dev, dev_index)
            # the same kernel with enough work per launch to leave the latency regime: a 1M-point source
            out["config3_with_1M_point_source"] = single_pair_rate(dev, dev_index, 3, max(5, a.steps // 5), 2, n_src=1_000_000)
            if not a.no_3d:
                out["scan_loop_3d"] = scan_loop_3d(dev, dev_index)
        copy_peak = stream_copy_GBps(dev)
        out["roofline"]["stream_copy_GBps"] = round(copy_peak, 1)
        out["roofline"]["frac_of_stream_copy"] = round(out["roofline"]["achieved"] / copy_peak, 4)
        if "batch" in out:
            out["batch"]["roofline"]["stream_copy_GBps"] = round(copy_peak, 1)
            out["batch"]["roofline"]["frac_of_stream_copy"] = round(out["batch"]["roofline"]["achieved"] / copy_peak, 4)
    else:
        # N > 1.  A single alignment is never split across GPUs (it would need an all-reduce per ~5 us iteration,
        # DESIGN.md section 8), so the line's `value` is the headline workload as ONE INDEPENDENT REPLICA PER GPU - the
        # same metric, unit and per-GPU work as the N = 1 line, comparable with it - and the path that does shard,
        # the loop-closure batch of config 4 (pairs split across ranks, one RCCL all_gather of the result rows), is
        # timed beside it under `batch`, comparable with the N = 1 line's `batch`.
        from gtsam_ndt_amd import dist as nd
        batch = run_batch(a, dev, dev_index, rank, world, dist, barrier)
        batch_conv = (run_batch(a, dev, dev_index, rank, world, dist, barrier, steps=max(3, min(a.steps, 10)), converged=True)
                      if a.converged_batch else None)
        d = synth.make_pair(3)
        tx, ty, sx, sy = (torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "sx", "sy"))
        torch.cuda.synchronize()
        m = NdtMatcher2D(device=dev_index, fixed_iterations=K_GN)
        m.set_target(tx, ty)
        n_src = int(sx.numel())

        def step():
            m.align_async(sx, sy, d["init"], producer_complete=True)

        for _ in range(a.warmup):
            step()
        m.finish()
        barrier()
        t0 = time.perf_counter()
        ev_ms = hip_events_ms(m.stream, lambda: [step() for _ in range(a.steps)])
        r = m.finish()
        barrier()
        elapsed = nd.max_over_ranks(time.perf_counter() - t0, device=dev)
        assert r.iterations == K_GN and r.status == 0
        m.close()
        launch_us = 1e3 * ev_ms / (a.steps * (K_GN + 1))
        alg_bytes = n_src * BYTES_PER_POINT_ITER
        achieved = alg_bytes / (launch_us * 1e-6) / 1e9
        out = {
            "metric": METRIC, "value": round(world * a.steps * K_GN / elapsed, 1), "unit": "iters/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_kind": "replicas (one independent config-3 alignment chain per GPU: linear by construction, no multi-GPU "
                          "code runs in its timed region; the sharded path with its RCCL gather is batch.value)",
            "config": {"workload": "config3: 2D NDT, 1M-pt submap target vs 100k-pt scan, single pair, 0.5 m cells, fixed 30 GN "
                                   "iterations per step - one independent replica per GPU (replicas only: a single alignment "
                                   "does not shard); the sharded loop-closure batch of config 4 is under `batch`",
                       "n_target": int(tx.numel()), "n_source": n_src, "cell_size": 0.5, "gn_iterations_per_step": K_GN,
                       "hessian": "gauss-newton", "replicas": world},
            "roofline": {"bound": "hbm", "kernel": "k_iterate<GN>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (load_traffic() or {}).get("bytes_per_launch"),
                         "traffic_source": "not measured in this run: committed PMC passes of the N = 1 command (profiles/pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(launch_us, 3),
                         "timing": "HIP events on rank 0's handle stream over the timed region / launches (per GPU)"},
            "scaling_note": "value: the N = 1 line's workload replicated per GPU (compare with the N = 1 line's value); "
                            "batch.value: the config-4 loop-closure batch sharded across the GPUs with its RCCL gather "
                            "(compare with the N = 1 line's batch.value)",
            "batch": batch,
        }
        if batch_conv is not None:
            out["batch_converged"] = batch_conv

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
