/* Driver for tests/test_oracle_sanitize.py: runs the C oracle (oracle/ndt_oracle.c, compiled together with this file
 * under -fsanitize=address,undefined) on a case file the test wrote, prints the result as text.
 *
 * Case file (little endian): int32 dim (2|3), int32 hessian_mode, int32 line_search, int32 threads, int32 fixed_iterations,
 * int32 pad, double step_scale, double cell_size, int32 min_points, int32 min_hits, uint64 n_target, uint64 n_source,
 * float target[dim][n_target], float source[dim][n_source], double init[3 or 6].
 * Test infrastructure only ("CPU code under -fsanitize=address,undefined in tests"). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_params {
  double cell_size; int32_t min_points, hessian_mode; double eig_ratio, d1, d2; int32_t max_iterations, fixed_iterations;
  double eps_trans, eps_rot, step_max_trans, step_max_rot; int32_t min_hits, reserved, line_search, reserved2; double step_scale;
} orc_params;
typedef struct orc_result { double pose[3], H[9], g[3], score; int32_t iterations, n_hit, status, reserved; } orc_result;
typedef struct orc_result3 { double pose[6], H[36], g[6], score; int32_t iterations, n_hit, status, reserved; } orc_result3;
typedef struct orc_grid2d orc_grid2d;
typedef struct orc_grid3d orc_grid3d;
orc_grid2d* orc2d_build_grid(const float*, const float*, size_t, const orc_params*);
void orc2d_free_grid(orc_grid2d*);
int32_t orc2d_align(const orc_grid2d*, const float*, const float*, size_t, const double*, const orc_params*, int, orc_result*);
orc_grid3d* orc3d_build_grid(const float*, const float*, const float*, size_t, const orc_params*);
void orc3d_free_grid(orc_grid3d*);
int32_t orc3d_align(const orc_grid3d*, const float*, const float*, const float*, size_t, const double*, const orc_params*, int,
                    orc_result3*);

static void need(int ok, const char* what) { if (!ok) { fprintf(stderr, "oracle_sanitize_main: %s\n", what); exit(2); } }

int main(int argc, char** argv) {
  need(argc == 2, "usage: oracle_sanitize_main <case file>");
  FILE* f = fopen(argv[1], "rb");
  need(f != NULL, "cannot open the case file");
  int32_t hd[6];
  double sc[2];
  int32_t mm[2];
  uint64_t n[2];
  need(fread(hd, sizeof(hd), 1, f) == 1 && fread(sc, sizeof(sc), 1, f) == 1 && fread(mm, sizeof(mm), 1, f) == 1 &&
       fread(n, sizeof(n), 1, f) == 1, "short header");
  const int dim = hd[0];
  need(dim == 2 || dim == 3, "dim must be 2 or 3");
  float* t = (float*)malloc(sizeof(float) * dim * (n[0] ? n[0] : 1));
  float* s = (float*)malloc(sizeof(float) * dim * (n[1] ? n[1] : 1));
  double init[6];
  need(fread(t, sizeof(float), dim * n[0], f) == dim * n[0] && fread(s, sizeof(float), dim * n[1], f) == dim * n[1] &&
       fread(init, sizeof(double), dim == 2 ? 3 : 6, f) == (size_t)(dim == 2 ? 3 : 6), "short body");
  fclose(f);
  orc_params p;
  memset(&p, 0, sizeof(p));
  p.cell_size = sc[1]; p.min_points = mm[0]; p.hessian_mode = hd[1]; p.eig_ratio = 1e-3; p.d1 = 1.0; p.d2 = 1.0;
  p.max_iterations = 100; p.fixed_iterations = hd[4]; p.eps_trans = 1e-5; p.eps_rot = 1e-5;
  p.step_max_trans = sc[1]; p.step_max_rot = 0.2; p.min_hits = mm[1]; p.line_search = hd[2]; p.step_scale = sc[0];
  if (dim == 2) {
    orc_grid2d* g = orc2d_build_grid(t, t + n[0], n[0], &p);
    need(g != NULL, "empty target");
    orc_result r;
    orc2d_align(g, s, s + n[1], n[1], init, &p, hd[3], &r);
    printf("%d %d %d %.17g %.17g %.17g %.17g\n", r.status, r.iterations, r.n_hit, r.pose[0], r.pose[1], r.pose[2], r.score);
    orc2d_free_grid(g);
  } else {
    orc_grid3d* g = orc3d_build_grid(t, t + n[0], t + 2 * n[0], n[0], &p);
    need(g != NULL, "empty target");
    orc_result3 r;
    orc3d_align(g, s, s + n[1], s + 2 * n[1], n[1], init, &p, hd[3], &r);
    printf("%d %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", r.status, r.iterations, r.n_hit, r.pose[0], r.pose[1], r.pose[2],
           r.pose[3], r.pose[4], r.pose[5], r.score);
    orc3d_free_grid(g);
  }
  free(t); free(s);
  return 0;
}
