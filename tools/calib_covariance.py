"""How the covariance calibration factors of ndt2d_calibrated_covariance (include/ndt_hip.h) were derived.

CPU Monte-Carlo with the C oracle (test infrastructure, like everything under oracle/): for a scene, a
point count and a noise level, R realisations of BOTH scans (fresh sampling and noise every time) are
aligned from the same initial guess; the empirical covariance C of the converged poses is compared
with S H^-1 S for the Gauss-Newton and the Newton form of H (eigenvalues of C (S H^-1 S)^-1, per
realisation; 1 = perfect).  The GPU test tests/test_gpu_covariance.py pins two of these cases.

    python tools/calib_covariance.py [kt_newton kr_newton kt_gn kr_gn]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gtsam_ndt_amd import synth
from oracle import cport, ndt2d as o


def pair(scene, n, seed_t, seed_s, pose, sigma):
    xt, yt = synth.sample_scene(scene, n, seed=seed_t, sigma=sigma)
    xs, ys = synth.sample_scene(scene, n, seed=seed_s, sigma=sigma)
    xs, ys = synth.to_source_frame(xs, ys, pose)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return f(xt), f(yt), f(xs), f(ys)


def main():
    a = [float(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [3.9, 7.5, 10.0, 18.0]
    KT, KR, GT, GR = a
    # (room size, points per scan, sigma, realisations, scene seed)
    cases = [(50.0, 100000, 0.03, 100, 2), (50.0, 20000, 0.03, 200, 2), (30.0, 20000, 0.03, 200, 7),
             (50.0, 20000, 0.01, 200, 2), (50.0, 5000, 0.03, 200, 2), (20.0, 10000, 0.02, 200, 11), (8.0, 3000, 0.03, 200, 1)]
    for (L, n, sigma, R, seed) in cases:
        scene = synth.room_scene(seed, L, -0.5 * L, -0.5 * L)
        pose = (0.10, -0.08, 0.01) if L > 10 else (0.05, -0.04, 0.01)
        prm = o.NdtParams()
        est, HN, HG = [], [], []
        for r in range(R):
            tx, ty, sx, sy = pair(scene, n, 1000 + 2 * r, 1001 + 2 * r, pose, sigma)
            g = cport.CGrid(tx, ty, prm)
            res = g.align(sx, sy, (0.0, 0.0, 0.0))
            if res["status"] == 0:
                est.append(res["pose"])
                if len(HN) < 30:
                    g1 = cport.CGrid(tx, ty, o.NdtParams(hessian_mode=1))
                    HN.append(g1.evaluate(sx, sy, res["pose"])[0])
                    g1.close()
                    HG.append(res["H"])
            g.close()
        C = np.cov(np.array(est).T)
        out = []
        for Hs, kt, kr in ((HN, KT, KR), (HG, GT, GR)):
            S = np.diag([np.sqrt(kt), np.sqrt(kt), np.sqrt(kr)])
            ev = np.array([np.sort(np.linalg.eigvals(np.linalg.solve(S @ np.linalg.inv(H) @ S, C)).real) for H in Hs])
            raw = np.array([np.sort(np.linalg.eigvals(np.linalg.solve(np.linalg.inv(H), C)).real) for H in Hs])
            out.append((ev.min(0), ev.max(0), raw.mean(0)))
        print(f"L={L} n={n} sigma={sigma} R={len(est)} emp std {np.sqrt(np.diag(C))}")
        for name, (lo, hi, raw) in zip(("Newton", "GN    "), out):
            print(f"   {name}-form: eig(C (S H^-1 S)^-1) in {np.round(lo, 2)} .. {np.round(hi, 2)};  eig(C H) = {np.round(raw, 1)}")


if __name__ == "__main__":
    main()
