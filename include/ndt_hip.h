/*
 * ndt_hip.h - C ABI of the MI355X (gfx950) NDT scan matcher.
 *
 * Drop-in boundary.  BASELINE.json's north_star names the interface
 * this library sits behind - "the repo's existing scan-matcher -> GTSAM-factor interface" -
 * but the reference checkout contains no header, class or signature for it:
 * /root/reference/README.md:1 ("# GTSAM-NDT") is the only line of the only file.  Each
 * entry point below therefore cites the SURVEY.md section 8 row it implements instead of a
 * reference file:line; INTEGRATION.md shows the adapter a maintainer of the reference
 * would write against this header once the real interface is visible.
 *
 * Conventions
 *   - extern "C", opaque handles, plain pointers and sizes, POD structs.  No C++ or torch
 *     types cross this boundary.
 *   - Every function returns an int32 status: 0 = NDT_OK; > 0 = numerical outcome of an
 *     alignment (also stored in ndt2d_result.status); < 0 = usage / HIP error.  No
 *     exception crosses the ABI; HIP errors are captured and returned.
 *   - Host-pointer entry points borrow the caller's buffers for the duration of the call
 *     and are synchronous.  "_dev" entry points take device pointers (any allocator:
 *     hipMalloc, torch) and enqueue on the given hipStream_t (passed as void*); results
 *     written to device memory are valid after that stream is synchronised.
 *   - A handle is single-threaded; distinct handles may be used from distinct threads.
 *   - Points are SoA float32 arrays (x[], y[]); poses are double (tx, ty, theta).
 *   - There is NO CPU fallback: every entry point that computes needs a gfx950 device.
 */
#ifndef NDT_HIP_H_
#define NDT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDT_ABI_VERSION 1

/* ---- status codes -------------------------------------------------------------- */
enum {
  NDT_OK = 0,
  NDT_NOT_CONVERGED = 1,      /* max_iterations reached                              */
  NDT_DEGENERATE_HESSIAN = 2, /* H + lambda*diag(H) never became positive definite    */
  NDT_TOO_FEW_HITS = 3,       /* fewer than min_hits source points fell in valid cells */
  NDT_TOO_FEW_CELLS = 4,      /* target grid has no valid cell                        */
  NDT_ERR_INVALID_ARG = -1,
  NDT_ERR_NO_TARGET = -2,
  NDT_ERR_HIP = -3,
  NDT_ERR_NO_DEVICE = -4,
  NDT_ERR_CAPACITY = -5,      /* a per-cell or per-grid capacity limit was exceeded   */
  NDT_ERR_ALLOC = -6,
  NDT_ERR_RCCL = -7           /* an RCCL call of the multi-device gather failed (ndt_last_error has its text) */
};

enum { NDT_HESSIAN_GAUSS_NEWTON = 0, NDT_HESSIAN_NEWTON = 1 };

/* ---- parameters (SURVEY.md section 5 "Config/flags": one POD at handle creation) -- */
typedef struct ndt2d_params {
  double cell_size;        /* c, metres                                   (row a1) */
  int32_t min_points;      /* a cell is valid iff n >= min_points          (row a3) */
  int32_t hessian_mode;    /* NDT_HESSIAN_*                                (row a6) */
  double eig_ratio;        /* lambda_min clamped to eig_ratio*lambda_max   (row a3) */
  double d1, d2;           /* score = sum d1*exp(-d2/2 * q' S^-1 q)        (row a5) */
  int32_t max_iterations;  /* cap on Gauss-Newton updates                  (row a8) */
  int32_t fixed_iterations;/* > 0: run exactly this many updates, no convergence test */
  double eps_trans;        /* converged when |dt| < eps_trans and ...      (row a8) */
  double eps_rot;          /* ... |dtheta| < eps_rot                               */
  double step_max_trans;   /* step is scaled so |dt| <= step_max_trans     (row a8) */
  double step_max_rot;     /* ... and |dtheta| <= step_max_rot                     */
  int32_t min_hits;        /* fewer hits than this => NDT_TOO_FEW_HITS             */
  int32_t overlap_grids;   /* 0 or 1: one grid; 4: Biber's four grids shifted by half a cell, every
                              point scores against all four (every 2D entry point: ndt2d_align*, the multi-start /
                              multi-scan chains, ndt2d_batch_*, ndt2d_multi_*)  */
  int32_t line_search;     /* 0: plain Gauss-Newton steps.  n in 1..16: backtracking line search - an
                              evaluation that scores worse than the pose its step started from (or
                              leaves the map) halves the step and retries from that pose, at most n
                              times per step; every trial is one evaluation and counts as one
                              iteration; convergence is tested on accepted evaluations only */
  int32_t reserved;
  double step_scale;       /* over-relaxation: the solved step is multiplied by this before the step
                              limits and the convergence test.  1 = plain Gauss-Newton/Newton.  The
                              Gauss-Newton Hessian of this score overestimates the true curvature
                              about 3x, so 2..3 cuts the iterations to
                              convergence 2..3x; keep 1 with NDT_HESSIAN_NEWTON.  0 means 1;
                              valid range (0, 8] */
} ndt2d_params;

/* ---- result (row a9) ------------------------------------------------------------ */
typedef struct ndt2d_result {
  double pose[3];     /* tx, ty, theta: maps source-frame points into the target frame */
  double H[9];        /* row-major 3x3 Hessian of -score at the last evaluated pose     */
  double g[3];        /* gradient of -score at the last evaluated pose                  */
  double score;       /* sum of per-point scores at the last evaluated pose             */
  int32_t iterations; /* Gauss-Newton updates applied                                   */
  int32_t n_hit;      /* source points that fell in a valid cell at the last evaluation */
  int32_t status;     /* NDT_OK / NDT_NOT_CONVERGED / ...                               */
  int32_t reserved;
} ndt2d_result;

/* one evaluation at a fixed pose (rows a4-a7), for stage parity tests and callers that
 * run their own optimiser */
typedef struct ndt2d_eval {
  double H[9];
  double g[3];
  double score;
  int32_t n_hit;
  int32_t reserved;
} ndt2d_eval;

/* geometry of the cached target grid (rows a1-a3) */
typedef struct ndt2d_grid_info {
  float ox, oy;       /* origin (lower-left corner of cell 0)  */
  float inv_cell;     /* float32(1/cell_size)                  */
  float cell;
  int32_t width, height;
  int32_t n_valid;    /* cells with n >= min_points and a usable covariance */
  int32_t n_points;   /* target points binned                                */
} ndt2d_grid_info;

typedef struct ndt2d_handle ndt2d_handle;

/* ---- lifecycle -------------------------------------------------------------------- */
int32_t ndt_abi_version(void);
const char* ndt_status_string(int32_t status);
/* last HIP error text captured on this thread ("" if none) */
const char* ndt_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int32_t ndt_device_count(void);
/* How a host thread waits for an alignment's end flag (pinned host memory).  0 (default): it spins on its core - a
 * call takes 60-300 us, less than a sleep's granularity.  1: it spins for the first ~20 us of a wait and then gives the
 * core away between polls (sched_yield) - for a process that drives several handles from more threads than it has
 * cores.  Process-wide, takes effect at the next wait; results do not depend on it. */
int32_t ndt_set_host_wait(int32_t mode);

void ndt2d_default_params(ndt2d_params* p);
int32_t ndt2d_create(const ndt2d_params* p, int32_t device_id, ndt2d_handle** out);
int32_t ndt2d_destroy(ndt2d_handle* h);

/* ---- (i) target voxel grid: rows a1-a3 ---------------------------------------------- */
/* Builds and caches the NDT grid of a target cloud. */
int32_t ndt2d_set_target(ndt2d_handle* h, const float* x, const float* y, size_t n);
int32_t ndt2d_set_target_dev(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n,
                             void* stream);
/* An empty grid over a caller-chosen extent (the region a submap is allowed to grow into),
 * filled afterwards with ndt2d_add_target_points(_dev).  The geometry is what ndt2d_set_target
 * derives from a cloud whose bounding box is [xmin, xmax] x [ymin, ymax] (as float32), so
 * reserve + add of a cloud's own bounding box gives bit for bit the grid of ndt2d_set_target.
 * Aligning against it before any cell is valid returns NDT_TOO_FEW_CELLS. */
int32_t ndt2d_reserve_target(ndt2d_handle* h, double xmin, double ymin, double xmax, double ymax);
/* Incremental submap update: bins n more points into the
 * cached grid's exact per-cell sums and re-finalises.  Points outside the cached extent
 * are counted in *n_outside (may be NULL) and ignored; so are points whose cell lies on the grid's
 * outermost ring (within one cell of the extent's border): the ring stays empty by contract,
 * because alignments clamp out-of-range lookups onto it. */
int32_t ndt2d_add_target_points(ndt2d_handle* h, const float* x, const float* y, size_t n,
                                size_t* n_outside);
/* The same with the points already on the device, optionally moved into the map frame first:
 * pose != NULL applies p' = R(theta) p + t in float32 (x' = (cs*x - sn*y) + tx, y' = (sn*x + cs*y)
 * + ty with cs, sn = (float)cos/sin(theta), every operation rounded separately) - the pose an
 * alignment of that scan returned - so a scan goes ranges -> points -> align -> submap without
 * leaving the GPU.  NaN points (no return) are ignored.  `stream` is the stream that produced
 * d_x/d_y (NULL: already complete); the call returns when the grid is updated. */
int32_t ndt2d_add_target_points_dev(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n,
                                    const double pose[3], size_t* n_outside, void* stream);
/* ---- submap persistence -------------------------------------------------------------------
 * The cached grid as a flat buffer: this header, then n_cells per-cell blocks of the EXACT fixed-point sums the
 * build keeps (2D, 48 B: int64 sx, sy, sxx, sxy, syy; uint32 n, pad.  3D, 80 B: int64 s[3], ss[6] (xx xy xz yy yz
 * zz); uint32 n, pad.  Coordinates are relative to the cell centre, in units of cell_size / 2^22).  Loading
 * re-finalises the sums with the LOADING handle's min_points / eig_ratio: with the saving handle's parameters the
 * grid is bit for bit the one that was saved, it goes on taking points (ndt2d_add_target_points*), and a map can be
 * re-finalised under other validity rules without the raw points.  cell_size (and overlap_grids) must match the
 * handle's (NDT_ERR_INVALID_ARG otherwise, as for a buffer that is not a map or is cut short); sums found in the
 * outermost ring of a 2D map, which the builders keep empty, are dropped on load.  The format is
 * little-endian, as the machines this library runs on. */
#define NDT_MAP_MAGIC 0x4d54444eu   /* "NDTM" */
typedef struct ndt_map_header {
  uint32_t magic, version;         /* NDT_MAP_MAGIC, 1 */
  int32_t dims, ngrid;             /* 2 or 3; 1, or 4 for the 2D overlapping grids (stored back to back) */
  int32_t width, height, depth;    /* cells per axis (depth 1 in 2D); x fastest, then y, then z */
  uint32_t cell_bytes;             /* 48 or 80 */
  double cell_size;
  uint64_t n_cells;                /* width * height * depth * ngrid */
  uint64_t n_points;               /* points binned so far (2D; 0 in 3D) */
  float origin[4][3];              /* lower corner of grid g */
} ndt_map_header;
/* bytes ndt2d_save_map writes for the cached grid (0 without one) */
size_t ndt2d_map_size(const ndt2d_handle* h);
/* *written (may be NULL) = the bytes needed, also when capacity is too small (NDT_ERR_CAPACITY) */
int32_t ndt2d_save_map(ndt2d_handle* h, void* buf, size_t capacity, size_t* written);
int32_t ndt2d_load_map(ndt2d_handle* h, const void* buf, size_t bytes);

int32_t ndt2d_get_grid_info(ndt2d_handle* h, ndt2d_grid_info* info);
/* Copies the finalised cell records to host arrays of width*height entries each
 * (any pointer may be NULL): count, mean (x,y interleaved), icov (a,b,c interleaved). */
int32_t ndt2d_get_grid(ndt2d_handle* h, int32_t* count, float* mean_xy, float* icov_abc);

/* ---- (ii)+(iii) evaluation at a fixed pose: rows a4-a7 ------------------------------- */
int32_t ndt2d_evaluate(ndt2d_handle* h, const float* sx, const float* sy, size_t n,
                       const double pose[3], ndt2d_eval* out);

/* the scan already on the device (ndt2d_wait_stream orders the handle behind its producer) */
int32_t ndt2d_evaluate_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n, const double pose[3],
                           ndt2d_eval* out);

/* ---- full alignment: rows a4-a9 ------------------------------------------------------ */
int32_t ndt2d_align(ndt2d_handle* h, const float* sx, const float* sy, size_t n,
                    const double init_pose[3], ndt2d_result* out);
/* Source already on the device.  Synchronous in the result (out is host memory). */
int32_t ndt2d_align_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                        const double init_pose[3], ndt2d_result* out);
/* Asynchronous form for timing and pipelining: returns as soon as the loop is under way on the
 * handle's stream; ndt2d_align_finish() waits and fetches.  With fixed_iterations > 0 the whole
 * loop is enqueued here; in converged mode the first launches are enqueued here and
 * ndt2d_align_finish() keeps the loop fed until it converges, so the d_sx / d_sy buffers must
 * stay valid until it returns (any other call on the handle finishes a loop in flight first). */
int32_t ndt2d_align_dev_async(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                              const double init_pose[3]);
int32_t ndt2d_align_finish(ndt2d_handle* h, ndt2d_result* out);
/* Multi-scan: m (1..64) DIFFERENT scans against the cached grid, each from its own initial pose, in one
 * launch chain - several robots', or several recent, scans relocalised in one submap.  d_sx / d_sy / n
 * are host arrays of m device pointers / sizes; results[k] is bit for bit what ndt2d_align_dev returns for
 * scan k on the launch-per-iteration path.  This is the call that takes the 1M-point-target configuration
 * off the launch-latency floor with distinct data: 64 scans of 100k points move 64 x 3.2 MB of algorithmic
 * traffic per launch. */
int32_t ndt2d_align_multi_scan_dev(ndt2d_handle* h, const float* const* d_sx, const float* const* d_sy, const size_t* n,
                                   const double* init_poses, int32_t m, ndt2d_result* results);
/* Per-iteration trace, for debugging and stage-by-stage parity checks ("optional
 * per-iteration trace ... copied back on request"; never on a timed path: one plain launch and one state
 * fetch per iteration, always through the launch-per-iteration kernels).  rows[j], j < *n_rows <= capacity,
 * is the state after j + 1 updates: pose after them, H / g / score / n_hit of the evaluation that
 * produced the (j+1)-th update (taken at rows[j-1].pose, the initial pose for j = 0), iterations = j + 1,
 * status.  out (may be NULL) receives the final result, as ndt2d_align returns it.  Host arrays. */
int32_t ndt2d_align_trace(ndt2d_handle* h, const float* sx, const float* sy, size_t n, const double init_pose[3],
                          ndt2d_result* rows, int32_t capacity, int32_t* n_rows, ndt2d_result* out);
/* Multi-start: m (1..64) independent alignments of the SAME scan against the cached grid from m
 * initial poses (init_poses is [m][3], host memory), carried by ONE launch chain - the source points
 * are read once per launch and every point scores against every live pose, so m alignments cost little
 * more than one (the single alignment is bound by launch latency, DESIGN.md section 5.1: 8 starts cost
 * 1.9x one, 64 starts 6x).  For a loop-closure candidate with a poor guess: a grid of starts around
 * it, keep the best score.
 * results[k] is bit for bit what ndt2d_align_dev returns for init_poses[k] on the
 * launch-per-iteration path; a start that has finished is frozen while the others go on.
 * Synchronous in the results (host memory).  overlap_grids = 4: the same chain with four lookups per point (one start per
 * workgroup), every start still bit for bit its single alignment with the option. */
int32_t ndt2d_align_multi_start_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                                    const double* init_poses, int32_t m, ndt2d_result* results);
/* Execution-strategy knobs of a handle.  They choose between kernels that compute the same alignment
 * (results agree up to float32 summation order; the tests pin each pair of choices against each other
 * and against the oracle); the defaults are the measured best, nothing reads the environment.
 *   NDT_TUNE_LAUNCH_GRAPHS      1 (default): launch chains are hipGraph replays; 0: plain launches
 *   NDT_TUNE_WIDE_THRESHOLD     source points from which k_iterate runs on 1024-thread workgroups
 *                               (default 300000); 0: never
 *   NDT_TUNE_SHORT_SCAN_KERNEL  1 (default): scans of <= 4096 points run the whole loop in one workgroup
 *   NDT_TUNE_CHUNK_LAUNCHES     converged mode: launches per graph replay, 2..128 (default 8)
 *   NDT_TUNE_BINNED_BUILD       1 (default): LDS-binned grid build; 0: scattered global atomics
 *   NDT_TUNE_SPLIT_FROM         multi-start / multi-scan calls of at least this many starts run two kernels per
 *                               iteration (one workgroup per start solves, then everybody evaluates) instead of
 *                               the fused kernel whose every workgroup repeats its starts' solves (default 12)
 *   NDT_TUNE_SINGLE_SYNC_BUILD  1 (default): ndt2d_set_target on a handle that already holds a grid decides the new grid's
 *                               geometry on the device and enqueues the whole build at once (one host round trip;
 *                               it falls back by itself when the new grid does not fit the cached storage).
 *                               0: bounding box to the host first, then the build (two round trips)
 *   NDT_TUNE_BATCH_SMALL_VARIANT (batch contexts) 1 (default): lidar-sized pairs run on the 256-thread
 *                               variant of the batch kernel first; 0: every pair on the 1024-thread one
 *   NDT_TUNE_BATCH_GLOBAL_WORKGROUPS (batch contexts, 2D and 3D) workgroups of the global-table variant, each with its own
 *                               table slab in device memory (3.7 MB in 2D, 7.9 MB in 3D): 1..256.  Unset, a context starts
 *                               with 8 (30 MB / 63 MB) and grows ONCE to one per CU (0.95 GB / 2.0 GB) at the start of the
 *                               first call after one in which a pair needed that variant - a context whose pairs all fit
 *                               on chip never pays for it; the call that first meets over-capacity pairs runs them on the
 *                               8.  Setting the knob fixes the number (64: a batch of over-capacity pairs about 2.8x
 *                               slower than on 256).  Results do not depend on it */
enum {
  NDT_TUNE_LAUNCH_GRAPHS = 1,
  NDT_TUNE_WIDE_THRESHOLD = 2,
  NDT_TUNE_SHORT_SCAN_KERNEL = 3,
  NDT_TUNE_CHUNK_LAUNCHES = 4,
  NDT_TUNE_BINNED_BUILD = 5,
  NDT_TUNE_BATCH_SMALL_VARIANT = 6,
  /* 7: was the one-XCD team kernel of round 2 (measured slower than the default, moved to tools/experiments) */
  NDT_TUNE_SPLIT_FROM = 8,
  NDT_TUNE_SINGLE_SYNC_BUILD = 9,
  NDT_TUNE_BATCH_GLOBAL_WORKGROUPS = 10
};
int32_t ndt2d_set_tuning(ndt2d_handle* h, int32_t knob, int64_t value);
/* hipStream_t the handle enqueues on (as void*), for event timing by the caller */
void* ndt2d_stream(ndt2d_handle* h);
/* Stream ordering of the device-pointer entry points.  A handle enqueues on its own non-blocking
 * stream, which is NOT ordered against the stream that produced the caller's device arrays (torch's
 * current stream, a driver's stream, the stream given to ndt2d_polar_to_points_dev).
 * ndt2d_wait_stream orders everything the handle enqueues after this call behind the work that is
 * in producer_stream now (event record + stream wait: the host does not block; NULL = the legacy
 * default stream).  Call it before ndt2d_align_dev / ndt2d_align_dev_async whenever the source
 * arrays were written on another stream and that stream has not been synchronised.
 * (ndt2d_set_target_dev and ndt2d_add_target_points_dev take the producer stream themselves.) */
int32_t ndt2d_wait_stream(ndt2d_handle* h, void* producer_stream);

/* ---- either side of the path --------------------------- */
/* Magnusson's outlier-mixture score constants (PhD thesis 2009, eq. 6.8-6.10) expressed as the
 * d1, d2 of ndt2d_params: with c1 = 10(1 - p_o), c2 = p_o / cell^dim, d3 = -ln c2,
 * d1 = -ln(c1 + c2) - d3, d2 = -2 ln((-ln(c1 e^-1/2 + c2) - d3) / d1).  The library maximises
 * sum d1' exp(-d2/2 m) with d1' = -d1 > 0, which is what is returned in *d1.  dim is 2 or 3. */
int32_t ndt_magnusson_constants(double outlier_ratio, double cell_size, int32_t dim, double* d1, double* d2);

/* Pose covariance for a factor-graph noise model from the Hessian an
 * alignment returned: cov = S H^-1 S with S = diag(s_t, s_t, s_r).  H^-1 itself is NOT the covariance
 * of the estimate: the score is a robust sum over cells, not a likelihood of independent points, and
 * its piecewise-smooth surface makes the fixed point of the iteration several times more sensitive to
 * the sampling of both scans than the local curvature says.  The factors are calibrated by Monte-Carlo
 * over noise realisations of both scans (tests/test_gpu_covariance.py):
 *   Gauss-Newton Hessian:  s_t^2 = 10,  s_r^2 = 18        Newton Hessian:  s_t^2 = 3.9,  s_r^2 = 7.5
 * and put the calibrated covariance within a factor 2.5 of the empirical one (every eigenvalue of
 * C_empirical C_calibrated^-1 in [0.4, 2.5]; the test asserts a factor 3) on scans with some 20 or more
 * points per occupied cell; the excess over H^-1 is discretisation noise and grows as the scans get
 * sparser (up to 5.5x off at 7 points per cell: inflate further there).  Needs no device.
 * Returns NDT_DEGENERATE_HESSIAN (cov zeroed) when H is not positive definite. */
#define NDT_COV_SCALE_GN_TRANS 10.0
#define NDT_COV_SCALE_GN_ROT 18.0
#define NDT_COV_SCALE_NEWTON_TRANS 3.9
#define NDT_COV_SCALE_NEWTON_ROT 7.5
int32_t ndt2d_calibrated_covariance(const double H[9], int32_t hessian_mode, double cov[9]);

/* Range/bearing scan -> SoA Cartesian points on the device (the driver side of the boundary):
 * x[i] = r[i] cos(angle_min + i*angle_inc), y likewise; ranges outside [range_min, range_max]
 * or non-finite become NaN points, which every entry point of this library ignores.
 * All pointers are device pointers; asynchronous on `stream` (NULL = default stream). */
int32_t ndt2d_polar_to_points_dev(const float* d_ranges, size_t n, double angle_min, double angle_inc,
                                  double range_min, double range_max, float* d_x, float* d_y, void* stream);

/* ---- loop-closure candidate batch (BASELINE config 4) -------------- */
/* Independent scan pairs, aligned concurrently: one persistent workgroup per CU pulls pairs
 * from a queue, builds the pair's target grid in LDS and runs its whole Gauss-Newton loop on
 * chip.  Clouds are concatenated SoA arrays; pair k owns target points [toff[k], toff[k+1])
 * and source points [soff[k], soff[k+1]); init is [n_pairs][3]; results is [n_pairs].
 * A pair whose grid exceeds the on-chip capacity (more than 20480 cells or 2559 occupied cells, e.g. a
 * scan against a submap wider than 71 m at 0.5 m cells) is handed, on the device and within the same
 * call, to a third variant of the kernel that keeps the pair's tables in global memory (up to 512 x 512
 * cells, 32767 occupied: 256 m x 256 m at 0.5 m cells).  Beyond that a pair gets status
 * NDT_ERR_CAPACITY from the _dev entry point, and the host-pointer entry point re-runs it through the
 * single-pair path transparently.
 * overlap_grids = 4: every pair of such a level runs on that third variant (four grids on chip would quarter
 * the capacity to 71 x 71 cells), its four grids back to back in the tables: 256 x 256 cells per grid
 * (128 m x 128 m at 0.5 m cells), 32767 occupied cells over the four; the context then holds one table slab
 * per CU from creation (0.95 GB).  Results equal ndt2d_align's with the same option up to float32 summation
 * order (tests/test_gpu_batch_overlap.py). */
typedef struct ndt2d_batch ndt2d_batch;
int32_t ndt2d_batch_create(const ndt2d_params* p, int32_t device_id, ndt2d_batch** out);
/* Coarse-to-fine over the batch (loop-closure candidates start from poor guesses): levels[0..n)
 * ordered coarse to fine, at most 8; every pair runs level 0 from its init and each later level
 * from the pose the previous one reached; a level that ends with a status other than NDT_OK /
 * NDT_NOT_CONVERGED ends the pair with that status; iterations are summed over levels and
 * H, g, score, n_hit are those of the last level run.  One kernel launch per level, the clouds
 * stay where they are.  ndt2d_default_pyramid fills the standard 3-level schedule (4c, 2c, c). */
int32_t ndt2d_default_pyramid(const ndt2d_params* fine, ndt2d_params levels[3]);
int32_t ndt2d_batch_create_pyramid(const ndt2d_params* levels, int32_t n_levels, int32_t device_id, ndt2d_batch** out);
int32_t ndt2d_batch_destroy(ndt2d_batch* b);
int32_t ndt2d_batch_align(ndt2d_batch* b, const float* tx, const float* ty, const uint64_t* toff,
                          const float* sx, const float* sy, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt2d_result* results);
/* All pointers are device pointers (results too).  Asynchronous on `stream` (NULL = the
 * context's own stream): the results are valid once that stream is synchronised.  Calls on one
 * context share its dequeue counters and pair marks, so they must be ordered with respect to each
 * other (the same stream, or synchronised streams); use one context per concurrent stream. */
int32_t ndt2d_batch_align_dev(ndt2d_batch* b, const float* d_tx, const float* d_ty, const uint64_t* d_toff,
                              const float* d_sx, const float* d_sy, const uint64_t* d_soff,
                              const double* d_init, size_t n_pairs, ndt2d_result* d_results, void* stream);
void* ndt2d_batch_stream(ndt2d_batch* b);
int32_t ndt2d_batch_set_tuning(ndt2d_batch* b, int32_t knob, int64_t value);   /* NDT_TUNE_BATCH_SMALL_VARIANT, _GLOBAL_WORKGROUPS */
/* as ndt2d_wait_stream, for calls that run on the context's own stream (stream == NULL above) */
int32_t ndt2d_batch_wait_stream(ndt2d_batch* b, void* producer_stream);
/* Of the pairs of the last ndt2d_batch_align() call, how many ran on the 1024-thread variant of
 * the kernel (the rest fitted the 256-thread variant for lidar-sized scans: clouds of at most 8192
 * points, at most 767 occupied cells); -1 before the first call.  Diagnostic. */
int64_t ndt2d_batch_last_large_count(const ndt2d_batch* b);

/* The same batch over several devices from ONE host process (a C++ SLAM back end that owns the
 * node's GPUs itself): one batch context and one host thread per device, pairs split into
 * contiguous work-balanced shards, results written into the caller's array.  Pairs are
 * independent, so no data moves between devices.  device_ids == NULL with n_devices == 0 means
 * every visible device; an id may be listed more than once (two contexts on that device; not with
 * the RCCL gather of ndt2d_multi_align_dev).  (The one-process-per-GPU deployment is
 * gtsam_ndt_amd/dist.py: the same sharding, the gather through torch.distributed's RCCL backend.) */
typedef struct ndt2d_multi ndt2d_multi;
int32_t ndt2d_multi_create(const ndt2d_params* p, const int32_t* device_ids, int32_t n_devices, ndt2d_multi** out);
int32_t ndt2d_multi_create_pyramid(const ndt2d_params* levels, int32_t n_levels, const int32_t* device_ids,
                                   int32_t n_devices, ndt2d_multi** out);
int32_t ndt2d_multi_destroy(ndt2d_multi* m);
int32_t ndt2d_multi_device_count(const ndt2d_multi* m);
/* Arguments as ndt2d_batch_align (host pointers).  Returns the first failing shard's status. */
int32_t ndt2d_multi_align(ndt2d_multi* m, const float* tx, const float* ty, const uint64_t* toff,
                          const float* sx, const float* sy, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt2d_result* results);
/* Device-resident form with the RCCL gather: shard d's clouds are already on device d (arrays of
 * n_devices device pointers, each laid out as ndt2d_batch_align_dev takes it; n_pairs[d] pairs on
 * device d, 0 allowed).  Every context aligns its shard on its own stream and the result rows of all
 * shards are exchanged with ONE ncclAllGather over those streams (xGMI on a multi-GPU node), so that
 * every device ends up holding every row - no result passes through host memory on the way.
 * Layout of a gathered copy: row k of shard d at index d * (*shard_stride) + k, *shard_stride = the
 * longest shard (shorter shards are zero-padded).  d_results_all[d] (may be NULL) receives device d's
 * copy - owned by the context, valid until its next call; results (may be NULL, host memory,
 * sum of n_pairs rows) receives the rows in global pair order without padding.  Synchronous.
 * The communicators are created on the first call (ncclCommInitAll: one process, one rank per
 * context, distinct devices required); RCCL failures return NDT_ERR_RCCL. */
int32_t ndt2d_multi_align_dev(ndt2d_multi* m, const float* const* d_tx, const float* const* d_ty,
                              const uint64_t* const* d_toff, const float* const* d_sx, const float* const* d_sy,
                              const uint64_t* const* d_soff, const double* const* d_init, const size_t* n_pairs,
                              ndt2d_result** d_results_all, size_t* shard_stride, ndt2d_result* results);
/* The split ndt2d_multi_align uses: shard d owns pairs [shard_begin[d], shard_begin[d+1]);
 * shard_begin has n_shards + 1 entries.  Work of a pair = 3 x target points + iterations_hint x
 * source points (iterations_hint <= 0: 30).  Needs no device. */
int32_t ndt2d_multi_plan(int32_t n_shards, const uint64_t* toff, const uint64_t* soff, size_t n_pairs,
                         int32_t iterations_hint, uint64_t* shard_begin);
/* The same split with a per-pair work hint (pairs of a converged-mode batch need different numbers
 * of iterations - 14 to 53 in this repository's tests): pair_iterations[k] > 0 replaces iterations_hint for pair k
 * (NULL: ndt2d_multi_plan).  For callers that place device-resident shards themselves (ndt2d_multi_align_dev). */
int32_t ndt2d_multi_plan_hinted(int32_t n_shards, const uint64_t* toff, const uint64_t* soff, size_t n_pairs,
                                int32_t iterations_hint, const int32_t* pair_iterations, uint64_t* shard_begin);

/* ---- 3D NDT, SE(3) (BASELINE config 5) ----------------------- */
/* Same pipeline in 3D: dense voxel grid with per-cell mean / 3x3 covariance (eigenvalue clamp
 * by a fixed-sweep Jacobi), pose = (tx, ty, tz, roll, pitch, yaw) with R = Rz(yaw) Ry(pitch)
 * Rx(roll), 6x6 Hessian (Gauss-Newton, or the full Newton form of Magnusson 2009 eq. 6.13 with
 * hessian_mode = NDT_HESSIAN_NEWTON), 6-vector gradient.  ndt3d_params has the layout and
 * meaning of ndt2d_params (overlap_grids is a 2D option). */
typedef ndt2d_params ndt3d_params;

typedef struct ndt3d_result {
  double pose[6];     /* tx ty tz roll pitch yaw                                  */
  double H[36];       /* row-major 6x6 Hessian (the form hessian_mode selects) at the last evaluation */
  double g[6];
  double score;
  int32_t iterations, n_hit, status, reserved;
} ndt3d_result;

typedef struct ndt3d_eval {
  double H[36];
  double g[6];
  double score;
  int32_t n_hit, reserved;
} ndt3d_eval;

typedef struct ndt3d_grid_info {
  float ox, oy, oz, inv_cell;
  int32_t width, height, depth;
  int32_t n_valid;
} ndt3d_grid_info;

typedef struct ndt3d_handle ndt3d_handle;
void ndt3d_default_params(ndt3d_params* p);   /* cell 1.0 m, min_points 5, step_max_trans 1.0, min_hits 6 */
int32_t ndt3d_create(const ndt3d_params* p, int32_t device_id, ndt3d_handle** out);
int32_t ndt3d_destroy(ndt3d_handle* h);
int32_t ndt3d_set_target(ndt3d_handle* h, const float* x, const float* y, const float* z, size_t n);
/* Incremental submap update, as ndt2d_add_target_points: bins n more points into the cached voxel
 * grid's exact sums and re-finalises; points outside the cached extent are counted and ignored. */
int32_t ndt3d_add_target_points(ndt3d_handle* h, const float* x, const float* y, const float* z, size_t n,
                                size_t* n_outside);
/* Range image of a spinning multi-beam lidar -> SoA Cartesian points on the device (the driver side of the 3D
 * boundary, as ndt2d_polar_to_points_dev): d_ranges is [n_elev][n_azim] row-major, n_elev <= 128; ring e looks up
 * at elevations[e] (host array, radians), column j at azimuth0 + j * azimuth_inc;
 * p = r (cos e cos a, cos e sin a, sin e).  Ranges outside [range_min, range_max] or non-finite become NaN points,
 * which every entry point of this library ignores.  d_* are device pointers; asynchronous on `stream`
 * (NULL = default stream). */
int32_t ndt3d_range_image_to_points_dev(const float* d_ranges, int32_t n_elev, int32_t n_azim, const double* elevations,
                                        double azimuth0, double azimuth_inc, double range_min, double range_max, float* d_x,
                                        float* d_y, float* d_z, void* stream);
/* Empty voxel grid over a chosen extent (lo / hi = min / max corner, x y z), to be filled with
 * ndt3d_add_target_points(_dev): a submap that grows scan by scan, as ndt2d_reserve_target. */
int32_t ndt3d_reserve_target(ndt3d_handle* h, const double lo[3], const double hi[3]);
/* ndt3d_add_target_points with the points already on the device, optionally moved into the map frame first:
 * pose != NULL applies p' = R p + t in float32 (R = Rz(yaw) Ry(pitch) Rx(roll) formed in float64 and rounded to
 * float32; each row ((r0 x + r1 y) + r2 z) + t with every operation rounded separately) - the pose an alignment
 * of that scan returned - so a 3D scan goes align -> submap without leaving the GPU.  `stream` is the stream that
 * produced the arrays (NULL: already complete); the call returns when the grid is updated. */
int32_t ndt3d_add_target_points_dev(ndt3d_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n,
                                    const double pose[6], size_t* n_outside, void* stream);
/* device arrays; `stream` = the stream that produced them (NULL: already complete) */
int32_t ndt3d_set_target_dev(ndt3d_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n,
                             void* stream);
/* submap persistence, as ndt2d_save_map / ndt2d_load_map (dims = 3, 80-byte cell blocks) */
size_t ndt3d_map_size(const ndt3d_handle* h);
int32_t ndt3d_save_map(ndt3d_handle* h, void* buf, size_t capacity, size_t* written);
int32_t ndt3d_load_map(ndt3d_handle* h, const void* buf, size_t bytes);
int32_t ndt3d_get_grid_info(ndt3d_handle* h, ndt3d_grid_info* info);
/* count [cells], mean [cells][3], icov [cells][6] (xx xy xz yy yz zz); any pointer may be NULL */
int32_t ndt3d_get_grid(ndt3d_handle* h, int32_t* count, float* mean_xyz, float* icov6);
int32_t ndt3d_evaluate(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                       const double pose[6], ndt3d_eval* out);
int32_t ndt3d_evaluate_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                           const double pose[6], ndt3d_eval* out);
int32_t ndt3d_align(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                    const double init_pose[6], ndt3d_result* out);
int32_t ndt3d_align_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                        const double init_pose[6], ndt3d_result* out);
/* Asynchronous form, as ndt2d_align_dev_async / ndt2d_align_finish: returns once the loop is under way
 * on the handle's stream (with fixed_iterations > 0 the whole chain and the fetch of its final state
 * are enqueued, so back-to-back calls keep the GPU busy without a host round trip per alignment);
 * the d_s* arrays must stay valid until ndt3d_align_finish returns.  Any other call on the handle
 * finishes an alignment in flight first. */
int32_t ndt3d_align_dev_async(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                              const double init_pose[6]);
int32_t ndt3d_align_finish(ndt3d_handle* h, ndt3d_result* out);
/* Many alignments against the cached voxel grid in ONE launch chain, as ndt2d_align_multi_scan_dev /
 * ndt2d_align_multi_start_dev: m (1..64) different scans, each from its own initial pose (d_sx / d_sy / d_sz / n:
 * host arrays of m device pointers / sizes; init_poses [m][6]) - or one scan from m initial poses.  Per
 * iteration the chain runs two kernels (one workgroup per start reduces and solves, then 256 x m workgroups
 * evaluate), so the launch boundary and the 6 x 6 solve are paid once per iteration for all starts.
 * results[k] is bit for bit what ndt3d_align_dev returns for scan k and init_poses[k]; a start that has
 * finished is frozen while the others go on.  Synchronous in the results (host memory). */
int32_t ndt3d_align_multi_scan_dev(ndt3d_handle* h, const float* const* d_sx, const float* const* d_sy, const float* const* d_sz,
                                   const size_t* n, const double* init_poses, int32_t m, ndt3d_result* results);
int32_t ndt3d_align_multi_start_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                                    const double* init_poses, int32_t m, ndt3d_result* results);
/* Per-iteration trace, as ndt2d_align_trace (debugging and stage-by-stage parity checks; never on a timed path):
 * rows[j], j < *n_rows <= capacity, is the state after j + 1 updates - the pose after them, H / g / score / n_hit
 * of the evaluation that produced the (j+1)-th update.  out (may be NULL): the final result.  Host arrays. */
int32_t ndt3d_align_trace(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                          const double init_pose[6], ndt3d_result* rows, int32_t capacity, int32_t* n_rows, ndt3d_result* out);
void* ndt3d_stream(ndt3d_handle* h);
/* as ndt2d_wait_stream: order the handle's stream behind the producer of the device arrays */
int32_t ndt3d_wait_stream(ndt3d_handle* h, void* producer_stream);
/* Execution-strategy knobs of a 3D handle (as ndt2d_set_tuning: they choose between code paths that build the same grid).
 *   NDT_TUNE_SINGLE_SYNC_BUILD  1 (default): ndt3d_set_target on a handle that already holds a grid decides the new grid's
 *                               geometry on the device and pays one host round trip; 0: the bounding box comes to the host
 *                               first (two round trips).  Other knobs: NDT_ERR_INVALID_ARG. */
int32_t ndt3d_set_tuning(ndt3d_handle* h, int32_t knob, int64_t value);

/* ---- 3D loop-closure candidate batch ------------------------------------------------------------ */
/* The 3D twin of ndt2d_batch: independent 3D scan pairs aligned concurrently, one persistent
 * 1024-thread workgroup per CU, the pair's voxel grid (u16 voxel -> slot table + 36-byte records) in
 * LDS for the whole Gauss-Newton / Newton loop.  Clouds are concatenated SoA arrays; pair k owns
 * target points [toff[k], toff[k+1]) and source points [soff[k], soff[k+1]); init is [n_pairs][6];
 * results is [n_pairs].  Every pair's result equals the single-pair path's (ndt3d_set_target +
 * ndt3d_align) up to float32 summation order: same records bit for bit, same per-point arithmetic.
 * On-chip capacity per pair: 2 B per voxel + 36 B per occupied voxel <= 157 KB and 6 B per voxel
 * <= 157 KB during the build (BASELINE config 5: 17 424 voxels, 2 706 occupied = 132 KB; that grid holds up to
 * 3 494 occupied voxels).  A pair
 * beyond it is handed, on the device and within the same call, to a second variant of the kernel that
 * keeps the pair's voxel table in global memory and its first 4 461 records on chip (up to 2^20 voxels - e.g.
 * 256 x 256 x 16 - and 32 767 occupied; a pair that only has a few more occupied voxels than fit stays on the first
 * variant with its last records in global memory).
 * Beyond that a pair gets status NDT_ERR_CAPACITY from the _dev entry point, and the host-pointer entry
 * point re-runs it through the single-pair path transparently.  Stream semantics as ndt2d_batch_align_dev.
 * A context holds about 0.2 GB of device memory (per-workgroup slabs of the build, 8 slabs of the global-memory variant)
 * and 2.1 GB once a call has needed the global-memory variant (NDT_TUNE_BATCH_GLOBAL_WORKGROUPS).
 * Results do not depend on the order of a cloud's points beyond float32 summation order (the grid not at all);
 * the grid build is fastest on scans left in the order a 64-beam driver delivers them (all beams of one bearing, then
 * the next bearing): it walks a cloud in rows of 64 points and combines a lane's consecutive rows in registers. */
typedef struct ndt3d_batch ndt3d_batch;
int32_t ndt3d_batch_create(const ndt3d_params* p, int32_t device_id, ndt3d_batch** out);
/* coarse-to-fine over the batch, as ndt2d_batch_create_pyramid (levels coarse to fine, at most 8) */
int32_t ndt3d_batch_create_pyramid(const ndt3d_params* levels, int32_t n_levels, int32_t device_id, ndt3d_batch** out);
int32_t ndt3d_batch_destroy(ndt3d_batch* b);
int32_t ndt3d_batch_align(ndt3d_batch* b, const float* tx, const float* ty, const float* tz, const uint64_t* toff,
                          const float* sx, const float* sy, const float* sz, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt3d_result* results);
int32_t ndt3d_batch_align_dev(ndt3d_batch* b, const float* d_tx, const float* d_ty, const float* d_tz, const uint64_t* d_toff,
                              const float* d_sx, const float* d_sy, const float* d_sz, const uint64_t* d_soff,
                              const double* d_init, size_t n_pairs, ndt3d_result* d_results, void* stream);
void* ndt3d_batch_stream(ndt3d_batch* b);
int32_t ndt3d_batch_wait_stream(ndt3d_batch* b, void* producer_stream);
int32_t ndt3d_batch_set_tuning(ndt3d_batch* b, int32_t knob, int64_t value);   /* NDT_TUNE_BATCH_GLOBAL_WORKGROUPS */

/* The 3D batch over several devices from ONE host process: ndt2d_multi_* for ndt3d_batch contexts (one context and one
 * host thread per device, contiguous work-balanced shards by ndt2d_multi_plan's rule).  ndt3d_multi_align takes host
 * pointers laid out as ndt3d_batch_align; ndt3d_multi_align_dev takes per-device device pointers (arrays of n_devices
 * pointers, shard d on device d, n_pairs[d] pairs, 0 allowed), aligns every shard on its context's stream and
 * exchanges the 408-byte result rows with ONE grouped ncclAllGather - layout, ownership and error codes as
 * ndt2d_multi_align_dev. */
typedef struct ndt3d_multi ndt3d_multi;
int32_t ndt3d_multi_create(const ndt3d_params* p, const int32_t* device_ids, int32_t n_devices, ndt3d_multi** out);
int32_t ndt3d_multi_create_pyramid(const ndt3d_params* levels, int32_t n_levels, const int32_t* device_ids,
                                   int32_t n_devices, ndt3d_multi** out);
int32_t ndt3d_multi_destroy(ndt3d_multi* m);
int32_t ndt3d_multi_device_count(const ndt3d_multi* m);
int32_t ndt3d_multi_align(ndt3d_multi* m, const float* tx, const float* ty, const float* tz, const uint64_t* toff,
                          const float* sx, const float* sy, const float* sz, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt3d_result* results);
int32_t ndt3d_multi_align_dev(ndt3d_multi* m, const float* const* d_tx, const float* const* d_ty, const float* const* d_tz,
                              const uint64_t* const* d_toff, const float* const* d_sx, const float* const* d_sy,
                              const float* const* d_sz, const uint64_t* const* d_soff, const double* const* d_init,
                              const size_t* n_pairs, ndt3d_result** d_results_all, size_t* shard_stride,
                              ndt3d_result* results);

#ifdef __cplusplus
}
#endif
#endif /* NDT_HIP_H_ */
