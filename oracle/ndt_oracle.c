/*
 * ndt_oracle.c - plain-C float64 restatement of oracle/ndt2d.py.  TEST INFRASTRUCTURE ONLY:
 * it is the checker and the timed "cpu_baseline" (kind "port") of bench.py, never a product
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference checkout holds no NDT source, test or golden vector
 * (/root/reference/README.md:1, "# GTSAM-NDT", is its only line), so this follows the
 * published algorithm (Biber & Strasser IROS 2003; Magnusson 2009; Welford 1962) with the
 * choices frozen in DESIGN.md section 2, exactly as oracle/ndt2d.py does.  tests/
 * test_oracle_c.py pins this file against the numpy oracle.
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
