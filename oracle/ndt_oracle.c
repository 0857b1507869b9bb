/*
 * ndt_oracle.c - plain-C float64 restatement of oracle/ndt2d.py (orc2d_*) and oracle/ndt3d.py
 * (orc3d_*, at the end of the file).  TEST INFRASTRUCTURE ONLY:
 * it is the checker and the timed "cpu_baseline" (kind "port") of bench.py, never a product
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference checkout holds no NDT source, test or golden vector
 * (/root/reference/README.md:1, "# GTSAM-NDT", is its only line), so this follows the
 * published algorithm (Biber & Strasser IROS 2003; Magnusson 2009; Welford 1962) with the
 * choices frozen in docs/ALGORITHM.md, exactly as oracle/ndt2d.py / ndt3d.py do.  tests/
 * test_oracle_c.py pins this file against the numpy oracles; tests/test_oracle_sanitize.py runs
 * it under -fsanitize=address,undefined.
 *
 * Built by gtsam_ndt_amd/build.py: gcc -O2 -std=c11 -fopenmp -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct orc_params {   /* same layout as ndt2d_params (include/ndt_hip.h) */
  double cell_size;
  int32_t min_points;
  int32_t hessian_mode;
  double eig_ratio;
  double d1, d2;
  int32_t max_iterations;
  int32_t fixed_iterations;
  double eps_trans, eps_rot;
  double step_max_trans, step_max_rot;
  int32_t min_hits;
  int32_t reserved;          /* overlap_grids: single grid only in this port */
  int32_t line_search;       /* > 0: backtracking, at most this many halvings per step */
  int32_t reserved2;
  double step_scale;         /* over-relaxation factor on the solved step (0 means 1) */
} orc_params;

typedef struct orc_result {   /* same layout as ndt2d_result */
  double pose[3];
  double H[9];
  double g[3];
  double score;
  int32_t iterations, n_hit, status, reserved;
} orc_result;

typedef struct orc_grid2d {
  float ox, oy, inv_c;
  int32_t W, H, n_valid;
  int64_t* count;
  double* mean;  /* [ncell][2] */
  double* icov;  /* [ncell][3] */
  uint8_t* valid;
} orc_grid2d;

/* ---- a3: oracle/ndt2d.py finalise_cell() ---------------------------------------------- */
static int finalise_cell(int64_t n, double m2xx, double m2xy, double m2yy, const orc_params* p,
                         double* a, double* b, double* c) {
  if (n < p->min_points || n < 2) return 0;
  const double sxx = m2xx / (double)(n - 1), sxy = m2xy / (double)(n - 1), syy = m2yy / (double)(n - 1);
  const double half_tr = 0.5 * (sxx + syy), half_df = 0.5 * (sxx - syy);
  const double disc = sqrt(half_df * half_df + sxy * sxy);
  const double l1 = half_tr + disc, l2 = half_tr - disc;
  if (!(l1 > 0.0)) return 0;
  const double lim = p->eig_ratio * l1;
  const double l2c = l2 > lim ? l2 : lim;
  double ex, ey;
  if (half_df >= 0.0) { ex = half_df + disc; ey = sxy; }
  else                { ex = sxy; ey = disc - half_df; }
  const double nrm = sqrt(ex * ex + ey * ey);
  if (nrm > 0.0) { ex /= nrm; ey /= nrm; } else { ex = 1.0; ey = 0.0; }
  const double i1 = 1.0 / l1, i2 = 1.0 / l2c, d = i1 - i2;
  *a = i2 + d * ex * ex;
  *b = d * ex * ey;
  *c = i2 + d * ey * ey;
  return 1;
}

void orc2d_free_grid(orc_grid2d* g) {
  if (!g) return;
  free(g->count); free(g->mean); free(g->icov); free(g->valid); free(g);
}

/* ---- a1-a3: oracle/ndt2d.py grid_geometry(), build_grid() -------------------------------- */
orc_grid2d* orc2d_build_grid(const float* x, const float* y, size_t n, const orc_params* p) {
  if (n == 0) return NULL;
  const double c = p->cell_size;
  float xmin = x[0], xmax = x[0], ymin = y[0], ymax = y[0];
  for (size_t i = 1; i < n; ++i) {
    if (x[i] < xmin) xmin = x[i];
    if (x[i] > xmax) xmax = x[i];
    if (y[i] < ymin) ymin = y[i];
    if (y[i] > ymax) ymax = y[i];
  }
  orc_grid2d* g = (orc_grid2d*)calloc(1, sizeof(*g));
  g->inv_c = (float)(1.0 / c);
  g->ox = (float)((floor((double)xmin / c) - 1.0) * c);
  g->oy = (float)((floor((double)ymin / c) - 1.0) * c);
  volatile float fxm = (xmax - g->ox) * g->inv_c, fym = (ymax - g->oy) * g->inv_c;
  g->W = (int32_t)floorf(fxm) + 2;
  g->H = (int32_t)floorf(fym) + 2;
  const size_t nc = (size_t)g->W * g->H;
  g->count = (int64_t*)calloc(nc, sizeof(int64_t));
  g->mean = (double*)calloc(nc * 2, sizeof(double));
  g->icov = (double*)calloc(nc * 3, sizeof(double));
  g->valid = (uint8_t*)calloc(nc, 1);
  int32_t* key = (int32_t*)malloc(n * sizeof(int32_t));
  double* mx = (double*)calloc(nc, sizeof(double));
  double* my = (double*)calloc(nc, sizeof(double));
  double* cx = (double*)calloc(nc, sizeof(double));
  double* cy = (double*)calloc(nc, sizeof(double));
  double* m2 = (double*)calloc(nc * 3, sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    volatile float fx = (x[i] - g->ox) * g->inv_c, fy = (y[i] - g->oy) * g->inv_c;
    /* a point whose cell lies on the outermost ring counts as outside (cell_keys32(interior=True)) */
    if (!(fx >= 1.0f && fx < (float)(g->W - 1) && fy >= 1.0f && fy < (float)(g->H - 1))) { key[i] = -1; continue; }
    const int32_t ix = (int32_t)floorf(fx), iy = (int32_t)floorf(fy);
    key[i] = iy * g->W + ix;
    g->count[key[i]] += 1;
    mx[key[i]] += (double)x[i];
    my[key[i]] += (double)y[i];
  }
  for (size_t k = 0; k < nc; ++k) {
    const double nz = g->count[k] > 0 ? (double)g->count[k] : 1.0;
    mx[k] /= nz; my[k] /= nz;
  }
  for (size_t i = 0; i < n; ++i) {   /* re-centre the mean */
    if (key[i] < 0) continue;
    cx[key[i]] += (double)x[i] - mx[key[i]];
    cy[key[i]] += (double)y[i] - my[key[i]];
  }
  for (size_t k = 0; k < nc; ++k) {
    const double nz = g->count[k] > 0 ? (double)g->count[k] : 1.0;
    mx[k] = mx[k] + cx[k] / nz;
    my[k] = my[k] + cy[k] / nz;
  }
  for (size_t i = 0; i < n; ++i) {
    if (key[i] < 0) continue;
    const double dx = (double)x[i] - mx[key[i]], dy = (double)y[i] - my[key[i]];
    m2[3 * (size_t)key[i]] += dx * dx;
    m2[3 * (size_t)key[i] + 1] += dx * dy;
    m2[3 * (size_t)key[i] + 2] += dy * dy;
  }
  for (size_t k = 0; k < nc; ++k) {
    double a, b, cc;
    if (finalise_cell(g->count[k], m2[3 * k], m2[3 * k + 1], m2[3 * k + 2], p, &a, &b, &cc)) {
      g->valid[k] = 1;
      g->n_valid += 1;
      g->mean[2 * k] = mx[k]; g->mean[2 * k + 1] = my[k];
      g->icov[3 * k] = a; g->icov[3 * k + 1] = b; g->icov[3 * k + 2] = cc;
    }
  }
  free(key); free(mx); free(my); free(cx); free(cy); free(m2);
  return g;
}

void orc2d_grid_info(const orc_grid2d* g, float* ox, float* oy, float* inv_c, int32_t* W, int32_t* H,
                     int32_t* n_valid) {
  *ox = g->ox; *oy = g->oy; *inv_c = g->inv_c; *W = g->W; *H = g->H; *n_valid = g->n_valid;
}

void orc2d_grid_copy(const orc_grid2d* g, int64_t* count, double* mean, double* icov, uint8_t* valid) {
  const size_t nc = (size_t)g->W * g->H;
  if (count) memcpy(count, g->count, nc * sizeof(int64_t));
  if (mean) memcpy(mean, g->mean, nc * 2 * sizeof(double));
  if (icov) memcpy(icov, g->icov, nc * 3 * sizeof(double));
  if (valid) memcpy(valid, g->valid, nc);
}

/* ---- a4-a7: oracle/ndt2d.py evaluate() (float64 truth mode) ------------------------------ */
void orc2d_evaluate(const orc_grid2d* g, const float* sx, const float* sy, size_t n, const double pose[3],
                    const orc_params* p, int threads, double H[9], double grad[3], double* score,
                    int32_t* n_hit) {
  const double tx = pose[0], ty = pose[1], cs = cos(pose[2]), sn = sin(pose[2]);
  const double ox = (double)g->ox, oy = (double)g->oy, inv_c = (double)g->inv_c;
  const double d1 = p->d1, d2 = p->d2;
  const int newton = p->hessian_mode == 1;
  double hxx = 0, hxy = 0, hyy = 0, hxt = 0, hyt = 0, htt = 0, gx = 0, gy = 0, gt = 0, sc = 0;
  long hits = 0;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static) \
    reduction(+ : hxx, hxy, hyy, hxt, hyt, htt, gx, gy, gt, sc, hits)
#endif
  for (long i = 0; i < (long)n; ++i) {
    const double x = (double)sx[i], y = (double)sy[i];
    const double px = cs * x - sn * y + tx, py = sn * x + cs * y + ty;
    const double fx = floor((px - ox) * inv_c), fy = floor((py - oy) * inv_c);
    if (!(fx >= 0.0 && fx < (double)g->W && fy >= 0.0 && fy < (double)g->H)) continue;
    const size_t k = (size_t)fy * g->W + (size_t)fx;
    if (!g->valid[k]) continue;
    const double qx = px - g->mean[2 * k], qy = py - g->mean[2 * k + 1];
    const double a = g->icov[3 * k], b = g->icov[3 * k + 1], c = g->icov[3 * k + 2];
    const double jx = -sn * x - cs * y, jy = cs * x - sn * y;
    const double vx = a * qx + b * qy, vy = b * qx + c * qy;
    const double m = qx * vx + qy * vy;
    const double s = d1 * exp(-0.5 * d2 * m);
    const double w = s * d2;
    const double vt = vx * jx + vy * jy;
    const double ux = a * jx + b * jy, uy = b * jx + c * jy;
    gx += w * vx; gy += w * vy; gt += w * vt;
    hxx += w * a; hxy += w * b; hyy += w * c;
    hxt += w * ux; hyt += w * uy; htt += w * (jx * ux + jy * uy);
    if (newton) {
      const double wd = w * d2;
      hxx -= wd * vx * vx; hxy -= wd * vx * vy; hyy -= wd * vy * vy;
      hxt -= wd * vx * vt; hyt -= wd * vy * vt; htt -= wd * vt * vt;
      htt += w * (vx * (-jy) + vy * jx);
    }
    sc += s;
    hits += 1;
  }
  H[0] = hxx; H[1] = hxy; H[2] = hxt;
  H[3] = hxy; H[4] = hyy; H[5] = hyt;
  H[6] = hxt; H[7] = hyt; H[8] = htt;
  grad[0] = gx; grad[1] = gy; grad[2] = gt;
  *score = sc;
  *n_hit = (int32_t)hits;
}

/* ---- a8: oracle/ndt2d.py solve3(), wrap_angle(), gn_update() ------------------------------ */
static int solve3(const double H[9], const double g[3], double d[3]) {
  const double h00 = H[0], h01 = H[1], h02 = H[2], h11 = H[4], h12 = H[5], h22 = H[8];
  const double d0 = fmax(fabs(h00), 1e-12), d1 = fmax(fabs(h11), 1e-12), d2 = fmax(fabs(h22), 1e-12);
  double lam = 0.0;
  for (int attempt = 0; attempt < 12; ++attempt) {
    const double a00 = h00 + lam * d0, a11 = h11 + lam * d1, a22 = h22 + lam * d2;
    if (a00 > 1e-12 * d0) {   /* LDL^T: pivots a00, p1, p2 */
      const double r0 = 1.0 / a00, l10 = h01 * r0, l20 = h02 * r0;
      const double p1 = a11 - l10 * h01;
      if (p1 > 1e-12 * d1) {
        const double r1 = 1.0 / p1, t = h12 - l20 * h01, l21 = t * r1;
        const double p2 = a22 - l20 * h02 - l21 * t;
        if (p2 > 1e-12 * d2) {
          const double r2 = 1.0 / p2;
          const double z0 = -g[0];
          const double z1 = -g[1] - l10 * z0;
          const double z2 = -g[2] - l20 * z0 - l21 * z1;
          const double x2 = z2 * r2;
          const double x1 = z1 * r1 - l21 * x2;
          const double x0 = z0 * r0 - l10 * x1 - l20 * x2;
          if (isfinite(x0) && isfinite(x1) && isfinite(x2)) { d[0] = x0; d[1] = x1; d[2] = x2; return 1; }
        }
      }
    }
    lam = lam == 0.0 ? 1e-6 : lam * 10.0;
  }
  return 0;
}

static double wrap_angle(double t) {
  const double pi = 3.141592653589793;
  if (t > pi || t <= -pi) {
    t = t - 2.0 * pi * floor((t + pi) / (2.0 * pi));
    if (t <= -pi) t += 2.0 * pi;
  }
  return t;
}

/* ---- a4-a9: oracle/ndt2d.py align() ------------------------------------------------------ */
int32_t orc2d_align(const orc_grid2d* g, const float* sx, const float* sy, size_t n, const double init[3],
                    const orc_params* p, int threads, orc_result* out) {
  memset(out, 0, sizeof(*out));
  double pose[3] = {init[0], init[1], init[2]};
  int it = 0, status = 0;
  if (g->n_valid < 1) {
    memcpy(out->pose, pose, sizeof(pose));
    out->status = 4;
    return 4;
  }
  /* backtracking line search state: oracle/ndt2d.py gn_update() */
  double ls_base[3] = {0, 0, 0}, ls_step[3] = {0, 0, 0}, ls_score = 0.0, ls_alpha = 1.0;
  int ls_valid = 0, ls_trials = 0;
  for (;;) {
    double d[3];
    orc2d_evaluate(g, sx, sy, n, pose, p, threads, out->H, out->g, &out->score, &out->n_hit);
    if (p->line_search > 0 && ls_valid && ls_trials < p->line_search &&
        (out->n_hit < p->min_hits || out->score < ls_score - 1e-3 * fabs(ls_score))) {
      ls_alpha *= 0.5;
      ls_trials += 1;
      pose[0] = ls_base[0] + ls_alpha * ls_step[0];
      pose[1] = ls_base[1] + ls_alpha * ls_step[1];
      pose[2] = wrap_angle(ls_base[2] + ls_alpha * ls_step[2]);
      it += 1;
      if (p->fixed_iterations > 0) { if (it >= p->fixed_iterations) break; continue; }
      if (it >= p->max_iterations) { status = 1; break; }
      continue;
    }
    if (out->n_hit < p->min_hits) { status = 3; break; }
    if (!solve3(out->H, out->g, d)) { status = 2; break; }
    { const double w = p->step_scale > 0.0 ? p->step_scale : 1.0; d[0] *= w; d[1] *= w; d[2] *= w; }
    const double nt = sqrt(d[0] * d[0] + d[1] * d[1]), nr = fabs(d[2]);
    double alpha = 1.0;
    if (nt > p->step_max_trans) alpha = p->step_max_trans / nt;
    if (nr * alpha > p->step_max_rot) alpha = p->step_max_rot / nr;
    if (p->line_search > 0) {
      for (int j = 0; j < 3; ++j) { ls_base[j] = pose[j]; ls_step[j] = d[j] * alpha; }
      ls_score = out->score; ls_alpha = 1.0; ls_trials = 0; ls_valid = 1;
    }
    pose[0] += d[0] * alpha;
    pose[1] += d[1] * alpha;
    pose[2] = wrap_angle(pose[2] + d[2] * alpha);
    it += 1;
    if (p->fixed_iterations > 0) { if (it >= p->fixed_iterations) break; continue; }
    if (nt * alpha < p->eps_trans && nr * alpha < p->eps_rot) break;
    if (it >= p->max_iterations) { status = 1; break; }
  }
  memcpy(out->pose, pose, sizeof(pose));
  out->iterations = it;
  out->status = status;
  return status;
}

int32_t orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}


/* ========================================================================================== */
/* 3D SE(3) variant: restatement of oracle/ndt3d.py.          */
/* Same parameter struct (ndt3d_params is ndt2d_params), result = ndt3d_result.              */
/* ========================================================================================== */

typedef struct orc_result3 {   /* same layout as ndt3d_result */
  double pose[6];
  double H[36];
  double g[6];
  double score;
  int32_t iterations, n_hit, status, reserved;
} orc_result3;

typedef struct orc_grid3d {
  float o[3], inv_c;
  int32_t dims[3], n_valid;
  int64_t* count;
  double* mean;  /* [ncell][3] */
  double* icov;  /* [ncell][6]  xx xy xz yy yz zz */
  uint8_t* valid;
} orc_grid3d;

#define ORC_JACOBI_SWEEPS 6

/* oracle/ndt3d.py jacobi_eig3(): cyclic Jacobi, fixed sweeps over (0,1), (0,2), (1,2) */
static void jacobi_eig3(double A[3][3], double V[3][3]) {
  static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < ORC_JACOBI_SWEEPS; ++sweep) {
    for (int e = 0; e < 3; ++e) {
      const int p = PQ[e][0], q = PQ[e][1], r = 3 - p - q;
      const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
      double t = 0.0;
      if (fabs(apq) > 1e-300) {
        const double tau = (aqq - app) / (2.0 * apq);
        t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));   /* tau^2 = inf -> t = 0 */
      }
      const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
      const double arp = A[r][p], arq = A[r][q];
      A[p][p] = app - t * apq;
      A[q][q] = aqq + t * apq;
      A[p][q] = A[q][p] = 0.0;
      A[r][p] = A[p][r] = c * arp - sn * arq;
      A[r][q] = A[q][r] = sn * arp + c * arq;
      for (int i = 0; i < 3; ++i) {
        const double vp = V[i][p], vq = V[i][q];
        V[i][p] = c * vp - sn * vq;
        V[i][q] = sn * vp + c * vq;
      }
    }
  }
}

/* oracle/ndt3d.py finalise_cells3() for one voxel: M2 = xx xy xz yy yz zz */
static int finalise_cell3(int64_t n, const double M2[6], const orc_params* p, double icov[6]) {
  const int64_t nmin = p->min_points > 2 ? p->min_points : 2;
  if (n < nmin) return 0;
  const double den = (double)(n - 1 > 1 ? n - 1 : 1);
  double A[3][3], V[3][3];
  A[0][0] = M2[0] / den; A[0][1] = A[1][0] = M2[1] / den; A[0][2] = A[2][0] = M2[2] / den;
  A[1][1] = M2[3] / den; A[1][2] = A[2][1] = M2[4] / den; A[2][2] = M2[5] / den;
  jacobi_eig3(A, V);
  const double lam[3] = {A[0][0], A[1][1], A[2][2]};
  const double lmax = fmax(lam[0], fmax(lam[1], lam[2]));
  if (!(lmax > 0.0)) return 0;
  double inv[3];
  for (int k = 0; k < 3; ++k) inv[k] = 1.0 / fmax(lam[k], p->eig_ratio * lmax);
  double C[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += V[i][k] * inv[k] * V[j][k];
      C[i][j] = a;
    }
  icov[0] = C[0][0]; icov[1] = C[0][1]; icov[2] = C[0][2]; icov[3] = C[1][1]; icov[4] = C[1][2]; icov[5] = C[2][2];
  return 1;
}

void orc3d_free_grid(orc_grid3d* g) {
  if (!g) return;
  free(g->count); free(g->mean); free(g->icov); free(g->valid); free(g);
}

/* oracle/ndt3d.py grid_geometry3(), cell_keys3(), build_grid3(): two-pass mean (sum, then re-centred), centred M2 */
orc_grid3d* orc3d_build_grid(const float* x, const float* y, const float* z, size_t n, const orc_params* p) {
  if (n == 0) return NULL;
  const float* P[3] = {x, y, z};
  const double c = p->cell_size;
  orc_grid3d* g = (orc_grid3d*)calloc(1, sizeof(*g));
  g->inv_c = (float)(1.0 / c);
  for (int a = 0; a < 3; ++a) {
    float mn = P[a][0], mx = P[a][0];
    for (size_t i = 1; i < n; ++i) { if (P[a][i] < mn) mn = P[a][i]; if (P[a][i] > mx) mx = P[a][i]; }
    g->o[a] = (float)((floor((double)mn / c) - 1.0) * c);
    volatile float f = (mx - g->o[a]) * g->inv_c;
    g->dims[a] = (int32_t)floorf(f) + 2;
  }
  const size_t nc = (size_t)g->dims[0] * g->dims[1] * g->dims[2];
  g->count = (int64_t*)calloc(nc, sizeof(int64_t));
  g->mean = (double*)calloc(nc * 3, sizeof(double));
  g->icov = (double*)calloc(nc * 6, sizeof(double));
  g->valid = (uint8_t*)calloc(nc, 1);
  int64_t* key = (int64_t*)malloc(n * sizeof(int64_t));
  double* mean = (double*)calloc(nc * 3, sizeof(double));
  double* corr = (double*)calloc(nc * 3, sizeof(double));
  double* m2 = (double*)calloc(nc * 6, sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    int64_t idx[3];
    int inside = 1;
    for (int a = 0; a < 3; ++a) {
      volatile float f = (P[a][i] - g->o[a]) * g->inv_c;
      idx[a] = (int64_t)floorf(f);
      if (idx[a] < 0 || idx[a] >= g->dims[a]) inside = 0;
    }
    key[i] = inside ? (idx[2] * g->dims[1] + idx[1]) * g->dims[0] + idx[0] : -1;
    if (!inside) continue;                /* cannot happen with this geometry (ndt3d.py asserts it) */
    g->count[key[i]] += 1;
    for (int a = 0; a < 3; ++a) mean[3 * key[i] + a] += (double)P[a][i];
  }
  for (size_t k = 0; k < nc; ++k) {
    const double nz = g->count[k] > 0 ? (double)g->count[k] : 1.0;
    for (int a = 0; a < 3; ++a) mean[3 * k + a] /= nz;
  }
  for (size_t i = 0; i < n; ++i)
    if (key[i] >= 0) for (int a = 0; a < 3; ++a) corr[3 * key[i] + a] += (double)P[a][i] - mean[3 * key[i] + a];
  for (size_t k = 0; k < nc; ++k) {
    const double nz = g->count[k] > 0 ? (double)g->count[k] : 1.0;
    for (int a = 0; a < 3; ++a) mean[3 * k + a] = mean[3 * k + a] + corr[3 * k + a] / nz;
  }
  for (size_t i = 0; i < n; ++i) {
    if (key[i] < 0) continue;
    const size_t k = (size_t)key[i];
    const double dx = (double)x[i] - mean[3 * k], dy = (double)y[i] - mean[3 * k + 1], dz = (double)z[i] - mean[3 * k + 2];
    m2[6 * k] += dx * dx; m2[6 * k + 1] += dx * dy; m2[6 * k + 2] += dx * dz;
    m2[6 * k + 3] += dy * dy; m2[6 * k + 4] += dy * dz; m2[6 * k + 5] += dz * dz;
  }
  for (size_t k = 0; k < nc; ++k) {
    double ic[6];
    if (finalise_cell3(g->count[k], m2 + 6 * k, p, ic)) {
      g->valid[k] = 1;
      g->n_valid += 1;
      for (int a = 0; a < 3; ++a) g->mean[3 * k + a] = mean[3 * k + a];
      for (int a = 0; a < 6; ++a) g->icov[6 * k + a] = ic[a];
    }
  }
  free(key); free(mean); free(corr); free(m2);
  return g;
}

void orc3d_grid_info(const orc_grid3d* g, float o[3], float* inv_c, int32_t dims[3], int32_t* n_valid) {
  for (int a = 0; a < 3; ++a) { o[a] = g->o[a]; dims[a] = g->dims[a]; }
  *inv_c = g->inv_c; *n_valid = g->n_valid;
}

void orc3d_grid_copy(const orc_grid3d* g, int64_t* count, double* mean, double* icov, uint8_t* valid) {
  const size_t nc = (size_t)g->dims[0] * g->dims[1] * g->dims[2];
  if (count) memcpy(count, g->count, nc * sizeof(int64_t));
  if (mean) memcpy(mean, g->mean, nc * 3 * sizeof(double));
  if (icov) memcpy(icov, g->icov, nc * 6 * sizeof(double));
  if (valid) memcpy(valid, g->valid, nc);
}

static void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}

/* oracle/ndt3d.py rot_and_derivs() + rot_second_derivs(): R = Rz Ry Rx; D[0..2] first derivatives (roll, pitch, yaw);
 * DD[0..5] second derivatives in the order (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) */
static void rot_derivs(double roll, double pitch, double yaw, double R[3][3], double D[3][3][3], double DD[6][3][3]) {
  const double ca = cos(roll), sa = sin(roll), cb = cos(pitch), sb = sin(pitch), cg = cos(yaw), sg = sin(yaw);
  const double Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
  const double Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}};
  const double Rz[3][3] = {{cg, -sg, 0}, {sg, cg, 0}, {0, 0, 1}};
  const double dRx[3][3] = {{0, 0, 0}, {0, -sa, -ca}, {0, ca, -sa}};
  const double dRy[3][3] = {{-sb, 0, cb}, {0, 0, 0}, {-cb, 0, -sb}};
  const double dRz[3][3] = {{-sg, -cg, 0}, {cg, -sg, 0}, {0, 0, 0}};
  const double ddRx[3][3] = {{0, 0, 0}, {0, -ca, sa}, {0, -sa, -ca}};
  const double ddRy[3][3] = {{-cb, 0, -sb}, {0, 0, 0}, {sb, 0, -cb}};
  const double ddRz[3][3] = {{-cg, sg, 0}, {-sg, -cg, 0}, {0, 0, 0}};
  double T[3][3];
#define ORC_PROD(Z, Y, X, OUT) do { mat3_mul(Z, Y, T); mat3_mul(T, X, OUT); } while (0)
  ORC_PROD(Rz, Ry, Rx, R);
  ORC_PROD(Rz, Ry, dRx, D[0]); ORC_PROD(Rz, dRy, Rx, D[1]); ORC_PROD(dRz, Ry, Rx, D[2]);
  ORC_PROD(Rz, Ry, ddRx, DD[0]); ORC_PROD(Rz, dRy, dRx, DD[1]); ORC_PROD(dRz, Ry, dRx, DD[2]);
  ORC_PROD(Rz, ddRy, Rx, DD[3]); ORC_PROD(dRz, dRy, Rx, DD[4]); ORC_PROD(ddRz, Ry, Rx, DD[5]);
#undef ORC_PROD
}

#define ORC3_NACC (36 + 6 + 9 + 1)   /* H (full 6x6), g, M = sum w v p' (Newton), score */

/* oracle/ndt3d.py evaluate3() in float64 truth mode.  Per-thread partial sums are added in thread order. */
void orc3d_evaluate(const orc_grid3d* g, const float* sx, const float* sy, const float* sz, size_t n, const double pose[6],
                    const orc_params* p, int threads, double H[36], double grad[6], double* score, int32_t* n_hit) {
  double R[3][3], D[3][3][3], DD[6][3][3];
  rot_derivs(pose[3], pose[4], pose[5], R, D, DD);
  const double t[3] = {pose[0], pose[1], pose[2]};
  const double o[3] = {(double)g->o[0], (double)g->o[1], (double)g->o[2]}, inv_c = (double)g->inv_c;
  const double d1 = p->d1, d2 = p->d2;
  const int newton = p->hessian_mode == 1;
  int nthr = threads > 0 ? threads : 1;
#ifndef _OPENMP
  nthr = 1;
#endif
  double* part = (double*)calloc((size_t)nthr * ORC3_NACC, sizeof(double));
  long* hits = (long*)calloc((size_t)nthr, sizeof(long));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthr)
#endif
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    const int tid = 0, nt = 1;
#endif
    double acc[ORC3_NACC];
    for (int k = 0; k < ORC3_NACC; ++k) acc[k] = 0.0;
    long nh = 0;
    const size_t lo = n * (size_t)tid / (size_t)nt, hi = n * (size_t)(tid + 1) / (size_t)nt;
    for (size_t i = lo; i < hi; ++i) {
      const double pt[3] = {(double)sx[i], (double)sy[i], (double)sz[i]};
      double pw[3];
      int64_t idx[3];
      int inside = 1;
      for (int a = 0; a < 3; ++a) {
        pw[a] = R[a][0] * pt[0] + R[a][1] * pt[1] + R[a][2] * pt[2] + t[a];
        const double f = floor((pw[a] - o[a]) * inv_c);
        if (!(f >= 0.0 && f < (double)g->dims[a])) inside = 0;
        idx[a] = (int64_t)f;
      }
      if (!inside) continue;
      const size_t k = (size_t)((idx[2] * g->dims[1] + idx[1]) * g->dims[0] + idx[0]);
      if (!g->valid[k]) continue;
      const double* ic = g->icov + 6 * k;
      const double C[3][3] = {{ic[0], ic[1], ic[2]}, {ic[1], ic[3], ic[4]}, {ic[2], ic[4], ic[5]}};
      const double q[3] = {pw[0] - g->mean[3 * k], pw[1] - g->mean[3 * k + 1], pw[2] - g->mean[3 * k + 2]};
      double v[3];
      for (int a = 0; a < 3; ++a) v[a] = C[a][0] * q[0] + C[a][1] * q[1] + C[a][2] * q[2];
      const double m = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
      const double s = d1 * exp(-0.5 * d2 * m), w = s * d2;
      double J[3][6];
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) J[a][b] = a == b ? 1.0 : 0.0;
        for (int r = 0; r < 3; ++r) J[a][3 + r] = D[r][a][0] * pt[0] + D[r][a][1] * pt[1] + D[r][a][2] * pt[2];
      }
      double CJ[3][6], Jv[6];
      for (int c6 = 0; c6 < 6; ++c6) {
        for (int a = 0; a < 3; ++a) CJ[a][c6] = C[a][0] * J[0][c6] + C[a][1] * J[1][c6] + C[a][2] * J[2][c6];
        Jv[c6] = J[0][c6] * v[0] + J[1][c6] * v[1] + J[2][c6] * v[2];
        acc[36 + c6] += w * Jv[c6];
      }
      for (int r = 0; r < 6; ++r)
        for (int c6 = 0; c6 < 6; ++c6) {
          double h = w * (J[0][r] * CJ[0][c6] + J[1][r] * CJ[1][c6] + J[2][r] * CJ[2][c6]);
          if (newton) h -= d2 * w * Jv[r] * Jv[c6];
          acc[6 * r + c6] += h;
        }
      if (newton)
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) acc[42 + 3 * a + b] += w * v[a] * pt[b];
      acc[51] += s;
      nh += 1;
    }
    memcpy(part + (size_t)tid * ORC3_NACC, acc, sizeof(acc));
    hits[tid] = nh;
  }
  double tot[ORC3_NACC];
  for (int k = 0; k < ORC3_NACC; ++k) tot[k] = 0.0;
  long nh = 0;
  for (int th = 0; th < nthr; ++th) {
    for (int k = 0; k < ORC3_NACC; ++k) tot[k] += part[(size_t)th * ORC3_NACC + k];
    nh += hits[th];
  }
  free(part); free(hits);
  memcpy(H, tot, 36 * sizeof(double));
  memcpy(grad, tot + 36, 6 * sizeof(double));
  if (newton) {   /* v' d2p'/dpk dpl on the rotation block: sum_ab Rkl[a][b] M[a][b] */
    static const int AB[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int e = 0; e < 6; ++e) {
      double t2 = 0.0;
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) t2 += DD[e][a][b] * tot[42 + 3 * a + b];
      const int a = AB[e][0], b = AB[e][1];
      H[6 * (3 + a) + 3 + b] += t2;
      if (a != b) H[6 * (3 + b) + 3 + a] += t2;
    }
  }
  *score = tot[51];
  *n_hit = (int32_t)nh;
}

/* oracle/ndt3d.py solve_ldl(), n = 6 */
static int solve_ldl6(const double H[36], const double g[6], double x[6]) {
  enum { N = 6 };
  double dg[N];
  for (int i = 0; i < N; ++i) dg[i] = fmax(fabs(H[N * i + i]), 1e-12);
  double lam = 0.0;
  for (int attempt = 0; attempt < 12; ++attempt) {
    double L[N][N], Dd[N];
    memset(L, 0, sizeof(L));
    int ok = 1;
    for (int j = 0; j < N && ok; ++j) {
      double pv = H[N * j + j] + lam * dg[j];
      for (int k = 0; k < j; ++k) pv -= L[j][k] * L[j][k] * Dd[k];
      if (!(pv > 1e-12 * dg[j])) { ok = 0; break; }
      Dd[j] = pv;
      for (int i = j + 1; i < N; ++i) {
        double a = H[N * i + j];
        for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k] * Dd[k];
        L[i][j] = a / pv;
      }
    }
    if (ok) {
      double zz[N];
      for (int i = 0; i < N; ++i) {
        double a = -g[i];
        for (int k = 0; k < i; ++k) a -= L[i][k] * zz[k];
        zz[i] = a;
      }
      int fin = 1;
      for (int i = N - 1; i >= 0; --i) {
        double a = zz[i] / Dd[i];
        for (int k = i + 1; k < N; ++k) a -= L[k][i] * x[k];
        x[i] = a;
      }
      for (int i = 0; i < N; ++i) if (!isfinite(x[i])) fin = 0;
      if (fin) return 1;
    }
    lam = lam == 0.0 ? 1e-6 : lam * 10.0;
  }
  for (int i = 0; i < N; ++i) x[i] = 0.0;
  return 0;
}

/* oracle/ndt3d.py align3() with gn_update3() inlined */
int32_t orc3d_align(const orc_grid3d* g, const float* sx, const float* sy, const float* sz, size_t n, const double init[6],
                    const orc_params* p, int threads, orc_result3* out) {
  memset(out, 0, sizeof(*out));
  double pose[6];
  memcpy(pose, init, sizeof(pose));
  int it = 0, status = 0;
  if (g->n_valid < 1) {
    memcpy(out->pose, pose, sizeof(pose));
    out->status = 4;
    return 4;
  }
  double ls_base[6] = {0}, ls_step[6] = {0}, ls_score = 0.0, ls_alpha = 1.0;
  int ls_valid = 0, ls_trials = 0;
  for (;;) {
    double d[6];
    orc3d_evaluate(g, sx, sy, sz, n, pose, p, threads, out->H, out->g, &out->score, &out->n_hit);
    if (p->line_search > 0 && ls_valid && ls_trials < p->line_search &&
        (out->n_hit < p->min_hits || out->score < ls_score - 1e-3 * fabs(ls_score))) {
      ls_alpha *= 0.5;
      ls_trials += 1;
      for (int j = 0; j < 3; ++j) pose[j] = ls_base[j] + ls_alpha * ls_step[j];
      for (int j = 3; j < 6; ++j) pose[j] = wrap_angle(ls_base[j] + ls_alpha * ls_step[j]);
      it += 1;
      if (p->fixed_iterations > 0) { if (it >= p->fixed_iterations) break; continue; }
      if (it >= p->max_iterations) { status = 1; break; }
      continue;
    }
    if (out->n_hit < p->min_hits) { status = 3; break; }
    if (!solve_ldl6(out->H, out->g, d)) { status = 2; break; }
    { const double w = p->step_scale > 0.0 ? p->step_scale : 1.0; for (int j = 0; j < 6; ++j) d[j] *= w; }
    const double nt = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), nr = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    double alpha = 1.0;
    if (nt > p->step_max_trans) alpha = p->step_max_trans / nt;
    if (nr * alpha > p->step_max_rot) alpha = p->step_max_rot / nr;
    for (int j = 0; j < 6; ++j) d[j] *= alpha;
    if (p->line_search > 0) {
      for (int j = 0; j < 6; ++j) { ls_base[j] = pose[j]; ls_step[j] = d[j]; }
      ls_score = out->score; ls_alpha = 1.0; ls_trials = 0; ls_valid = 1;
    }
    for (int j = 0; j < 3; ++j) pose[j] += d[j];
    for (int j = 3; j < 6; ++j) pose[j] = wrap_angle(pose[j] + d[j]);
    it += 1;
    if (p->fixed_iterations > 0) { if (it >= p->fixed_iterations) break; continue; }
    if (nt * alpha < p->eps_trans && nr * alpha < p->eps_rot) break;
    if (it >= p->max_iterations) { status = 1; break; }
  }
  memcpy(out->pose, pose, sizeof(pose));
  out->iterations = it;
  out->status = status;
  return status;
}
