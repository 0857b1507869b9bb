"""Monte-Carlo behind the 3D covariance calibration (ndt3d_calibrated_covariance): for a scene and a fixed relative pose,
R realisations of BOTH scans - the pair is placed at a random pose in the room every time, so the beams sample
different surface points, and the range noise is fresh - are aligned from the same initial guess; the empirical
covariance C of the estimated relative pose is compared with S H^-1 S for the Gauss-Newton and the Newton form of H
(eigenvalues of C (S H^-1 S)^-1 per realisation; 1 = perfect).  Scans come from the device generator.

    python tools/calib_covariance3d.py [s_t2_gn s_r2_gn s_t2_newton s_r2_newton]
"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gtsam_ndt_amd import synth3d, synth_dev
from gtsam_ndt_amd.matcher import NdtMatcher3D


def euler(R):
    return math.atan2(R[2, 1], R[2, 2]), -math.asin(R[2, 0]), math.atan2(R[1, 0], R[0, 0])


def main():
    a = [float(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [1.0, 1.0, 1.0, 1.0]
    rel = tuple(0.5 * np.array(synth3d.T_STAR_3D))
    R_rel, t_rel = synth3d.rotation(*rel[3:]), np.array(rel[:3])
    for scene, shape, sigma, n_real in ((5, (64, 2048), 0.02, 300), (8, (64, 2048), 0.02, 300), (11, (32, 1024), 0.02, 300),
                                        (5, (64, 2048), 0.05, 300), (13, (16, 512), 0.02, 300)):
        rng = np.random.default_rng(scene)
        est, HG, HN = [], [], []
        with NdtMatcher3D() as m, NdtMatcher3D(hessian_mode=1, fixed_iterations=1) as mn:
            for r in range(n_real):
                A = (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01),
                     rng.uniform(-0.01, 0.01), rng.uniform(-0.3, 0.3))
                RA = synth3d.rotation(*A[3:])
                RB = RA @ R_rel
                B = tuple(np.array(A[:3]) + RA @ t_rel) + euler(RB)
                t = synth_dev.lidar_scan3d(5000 + 2 * r, A, shape[0], shape[1], sigma, scene_seed=scene)
                s = synth_dev.lidar_scan3d(5001 + 2 * r, B, shape[0], shape[1], sigma, scene_seed=scene)
                m.set_target(*t)
                res = m.align(*s, (0.0,) * 6)
                if res.status != 0:
                    continue
                est.append(res.pose)
                if len(HG) < 40:
                    HG.append(res.H)
                    mn.set_target(*t)
                    HN.append(mn.align(*s, res.pose).H)            # Newton form of H at the converged pose
        est = np.array(est)
        C = np.cov(est.T)
        bias = est.mean(0) - np.array(rel)
        print(f"scene {scene} {shape[0]}x{shape[1]} sigma {sigma}: {len(est)} of {n_real} converged; emp std t {np.sqrt(np.diag(C))[:3].round(5)} "
              f"r {np.sqrt(np.diag(C))[3:].round(6)}; bias {np.abs(bias).max():.5f}")
        for name, Hs, kt, kr in (("GN    ", HG, a[0], a[1]), ("Newton", HN, a[2], a[3])):
            S = np.diag([math.sqrt(kt)] * 3 + [math.sqrt(kr)] * 3)
            ev, raw_t, raw_r = [], [], []
            for H in Hs:
                try:
                    Hi = np.linalg.inv(H)
                except np.linalg.LinAlgError:
                    continue
                if np.any(np.linalg.eigvalsh(0.5 * (H + H.T)) <= 0):
                    continue
                ev.append(np.sort(np.linalg.eigvals(np.linalg.solve(S @ Hi @ S, C)).real))
                raw_t.append(np.sort(np.linalg.eigvals(np.linalg.solve(Hi[:3, :3], C[:3, :3])).real))
                raw_r.append(np.sort(np.linalg.eigvals(np.linalg.solve(Hi[3:, 3:], C[3:, 3:])).real))
            ev = np.array(ev)
            print(f"   {name}: usable H {len(ev)}; eig(C (S H^-1 S)^-1) min {ev.min(0).round(2)} max {ev.max(0).round(2)}; "
                  f"block ratios C_tt / (H^-1)_tt {np.mean(raw_t, 0).round(1)}  C_rr / (H^-1)_rr {np.mean(raw_r, 0).round(1)}")


if __name__ == "__main__":
    main()
