// Submap persistence (included at the end of ndt2d_api.hip: one translation unit): the cached grid leaves a
// handle as its geometry plus the EXACT per-cell fixed-point sums the build keeps (CellAcc / CellAcc3), and comes
// back by re-finalising those sums with the loading handle's min_points / eig_ratio - so a reloaded submap is bit
// for bit the grid that was saved (same parameters), goes on taking points (ndt*_add_target_points*), and can be
// re-finalised under other validity rules without the raw points.  A front end can drop a submap from device
// memory and bring it back, or write it to disk, at 48 B (2D) / 80 B (3D) per cell.
#pragma once
#include <cmath>
#include <cstring>

static_assert(sizeof(ndt_map_header) == 104, "ndt_map_header is part of the ABI");
static_assert(sizeof(CellAcc) == 48 && sizeof(ndt::CellAcc3) == 80, "per-cell blocks of the map format");

namespace {

constexpr uint32_t kMapVersion = 1;

// The outermost ring of a 2D grid stays empty by contract (alignments clamp out-of-range lookups onto it,
// ndt2d_add_target_points never fills it): a map that did not come from ndt2d_save_map must not break that.
__global__ void k_clear_ring(CellAcc* __restrict__ acc, int W, int H, int ngrid) {
  const int per = 2 * W + 2 * H;                     // (corners twice: harmless)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * ngrid) return;
  const int q = i / per, j = i - q * per;
  int ix, iy;
  if (j < W) { ix = j; iy = 0; }
  else if (j < 2 * W) { ix = j - W; iy = H - 1; }
  else if (j < 2 * W + H) { ix = 0; iy = j - 2 * W; }
  else { ix = W - 1; iy = j - 2 * W - H; }
  acc[(size_t)q * W * H + (size_t)iy * W + ix] = CellAcc{};
}

int32_t check_map_header(const ndt_map_header& m, size_t bytes, int dims, uint32_t cell_bytes, double cell_size) {
  if (m.magic != NDT_MAP_MAGIC || m.version != kMapVersion) { set_error("not an NDT map of this format version"); return NDT_ERR_INVALID_ARG; }
  if (m.dims != dims || m.cell_bytes != cell_bytes) { set_error("map has another dimension"); return NDT_ERR_INVALID_ARG; }
  if (m.width < 1 || m.height < 1 || m.depth < 1 || (m.ngrid != 1 && m.ngrid != 4)) { set_error("map header: extents"); return NDT_ERR_INVALID_ARG; }
  if (dims == 2 && m.depth != 1) { set_error("map header: a 2D map has depth 1"); return NDT_ERR_INVALID_ARG; }
  for (int q = 0; q < (m.ngrid == 4 ? 4 : 1); ++q)
    for (int a = 0; a < dims; ++a)
      if (!std::isfinite(m.origin[q][a])) { set_error("map header: origin is not finite"); return NDT_ERR_INVALID_ARG; }
  const double nc = (double)m.width * m.height * m.depth * m.ngrid;
  if (nc > (double)kMaxCells || (uint64_t)nc != m.n_cells) { set_error("map header: cell count"); return NDT_ERR_INVALID_ARG; }
  if (bytes < sizeof(ndt_map_header) + (size_t)m.n_cells * cell_bytes) { set_error("map buffer is shorter than its header says"); return NDT_ERR_INVALID_ARG; }
  if (m.cell_size != cell_size) { set_error("map was built with another cell_size than this handle's"); return NDT_ERR_INVALID_ARG; }
  return NDT_OK;
}

// The per-cell blocks of a map are trusted by the finalise kernels (they are the library's own exact sums), so a
// buffer that did not come from ndt*_save_map is checked on the host before it is uploaded: a cell of n points holds
// fixed-point coordinates |U| <= 2^21 (checked against 2^22: the rounding of a boundary point), so
// |sum U| <= n 2^22, 0 <= sum U^2 <= n 2^44, |sum UV| <= n 2^44, n <= 2^20 (the capacity the sums are exact for), and
// the variance numerators n sum U^2 - (sum U)^2 are not negative (Cauchy-Schwarz; 128-bit exact).  Sums inside these
// bounds finalise into finite records whatever else they are; outside them the map is refused.
template <int DIM>
bool cell_sums_plausible(unsigned int n, const long long* s, const long long* ss_diag, const long long* ss_off, int n_off) {
  if (n > kMaxCellCount) return false;
  const __int128 nn = n, lim1 = nn << 22, lim2 = nn << 44;
  for (int a = 0; a < DIM; ++a) {
    const __int128 su = s[a], suu = ss_diag[a];
    if (su > lim1 || su < -lim1 || suu < 0 || suu > lim2) return false;
    if (nn * suu - su * su < 0) return false;
  }
  for (int a = 0; a < n_off; ++a)
    if ((__int128)ss_off[a] > lim2 || (__int128)ss_off[a] < -lim2) return false;
  return true;
}

int32_t check_map_cells2(const void* cells, size_t n_cells) {
  const char* p = static_cast<const char*>(cells);
  for (size_t k = 0; k < n_cells; ++k, p += sizeof(CellAcc)) {
    CellAcc c;
    std::memcpy(&c, p, sizeof c);
    const long long s[2] = {c.sx, c.sy}, dg[2] = {c.sxx, c.syy}, off[1] = {c.sxy};
    if (!cell_sums_plausible<2>(c.n, s, dg, off, 1)) { set_error("map cell sums are not those of any point set (forged or damaged map)"); return NDT_ERR_INVALID_ARG; }
  }
  return NDT_OK;
}

int32_t check_map_cells3(const void* cells, size_t n_cells) {
  const char* p = static_cast<const char*>(cells);
  for (size_t k = 0; k < n_cells; ++k, p += sizeof(ndt::CellAcc3)) {
    ndt::CellAcc3 c;
    std::memcpy(&c, p, sizeof c);
    const long long dg[3] = {c.ss[0], c.ss[3], c.ss[5]}, off[3] = {c.ss[1], c.ss[2], c.ss[4]};
    if (!cell_sums_plausible<3>(c.n, c.s, dg, off, 3)) { set_error("map cell sums are not those of any point set (forged or damaged map)"); return NDT_ERR_INVALID_ARG; }
  }
  return NDT_OK;
}

}  // namespace

extern "C" {

size_t ndt2d_map_size(const ndt2d_handle* h) {
  if (!h || !h->has_target) return 0;
  return sizeof(ndt_map_header) + (size_t)h->grid.W * h->grid.H * h->grid.ngrid * sizeof(CellAcc);
}

int32_t ndt2d_save_map(ndt2d_handle* h, void* buf, size_t capacity, size_t* written) {
  if (!h || !buf) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  const size_t need = ndt2d_map_size(h);
  if (written) *written = need;
  if (capacity < need) return NDT_ERR_CAPACITY;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  const GridDev& g = h->grid;
  ndt_map_header m{};
  m.magic = NDT_MAP_MAGIC; m.version = kMapVersion;
  m.dims = 2; m.ngrid = g.ngrid;
  m.width = g.W; m.height = g.H; m.depth = 1;
  m.cell_bytes = (uint32_t)sizeof(CellAcc);
  m.cell_size = g.cell;
  m.n_cells = (uint64_t)g.W * g.H * g.ngrid;
  m.n_points = (uint64_t)h->n_points;
  for (int q = 0; q < 4; ++q) { m.origin[q][0] = g.gx[q]; m.origin[q][1] = g.gy[q]; m.origin[q][2] = 0.f; }
  std::memcpy(buf, &m, sizeof m);
  HIP_TRY(hipMemcpyAsync((char*)buf + sizeof m, g.acc, (size_t)m.n_cells * sizeof(CellAcc), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return NDT_OK;
}

int32_t ndt2d_load_map(ndt2d_handle* h, const void* buf, size_t bytes) {
  if (!h || !buf || bytes < sizeof(ndt_map_header)) return NDT_ERR_INVALID_ARG;
  ndt_map_header m;
  std::memcpy(&m, buf, sizeof m);
  { const int32_t cs = check_map_header(m, bytes, 2, (uint32_t)sizeof(CellAcc), h->prm.cell_size); if (cs != NDT_OK) return cs; }
  if (m.ngrid != (h->prm.overlap_grids == 4 ? 4 : 1)) { set_error("map and handle differ in overlap_grids"); return NDT_ERR_INVALID_ARG; }
  { const int32_t cs = check_map_cells2((const char*)buf + sizeof m, (size_t)m.n_cells); if (cs != NDT_OK) return cs; }
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  h->has_target = false;
  GridDev& g = h->grid;
  const double c = h->prm.cell_size;
  g.cell = c; g.cell32 = (float)c; g.inv_c = (float)(1.0 / c);
  g.W = m.width; g.H = m.height; g.ngrid = m.ngrid; g.pad = 0;
  for (int q = 0; q < kMaxGrids; ++q) { g.gx[q] = m.origin[q][0]; g.gy[q] = m.origin[q][1]; }
  g.ox = g.gx[0]; g.oy = g.gy[0];
  g.fix_scale = std::ldexp(1.0, kFixShift) / c;
  const size_t ncell = (size_t)m.n_cells;
  if (ncell > h->cell_capacity) {
    if (g.rec) (void)hipFree(g.rec);
    if (g.acc) (void)hipFree(g.acc);
    g.rec = nullptr; g.acc = nullptr; h->cell_capacity = 0;
    const size_t want = ncell + ncell / 8;
    HIP_TRY(hipMalloc((void**)&g.rec, 2 * want * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&g.acc, want * sizeof(CellAcc)));
    h->cell_capacity = want;
  }
  HIP_TRY(hipMemcpyAsync(g.acc, (const char*)buf + sizeof m, ncell * sizeof(CellAcc), hipMemcpyHostToDevice, h->stream));
  {
    const int n_ring = (2 * g.W + 2 * g.H) * g.ngrid;
    hipLaunchKernelGGL(k_clear_ring, dim3((unsigned)((n_ring + 255) / 256)), dim3(256), 0, h->stream, g.acc, g.W, g.H, g.ngrid);
    HIP_TRY(hipGetLastError());
  }
  { const int32_t fs = finalise_grid(h); if (fs != NDT_OK) return fs; }      // synchronises: buf is free on return
  h->n_points = (size_t)m.n_points;
  h->last_ntile = 0;                 // the next ndt2d_set_target sizes its launches the two-round-trip way once
  h->has_target = true;
  return upload_static(h);
}

size_t ndt3d_map_size(const ndt3d_handle* h) {
  if (!h || !h->has_target) return 0;
  return sizeof(ndt_map_header) + (size_t)h->grid.W * h->grid.H * h->grid.D * sizeof(ndt::CellAcc3);
}

int32_t ndt3d_save_map(ndt3d_handle* h, void* buf, size_t capacity, size_t* written) {
  if (!h || !buf) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  const size_t need = ndt3d_map_size(h);
  if (written) *written = need;
  if (capacity < need) return NDT_ERR_CAPACITY;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  const ndt::Grid3Dev& g = h->grid;
  ndt_map_header m{};
  m.magic = NDT_MAP_MAGIC; m.version = kMapVersion;
  m.dims = 3; m.ngrid = 1;
  m.width = g.W; m.height = g.H; m.depth = g.D;
  m.cell_bytes = (uint32_t)sizeof(ndt::CellAcc3);
  m.cell_size = g.cell;
  m.n_cells = (uint64_t)g.W * g.H * g.D;
  m.n_points = 0;                    // not tracked in 3D
  m.origin[0][0] = g.ox; m.origin[0][1] = g.oy; m.origin[0][2] = g.oz;
  std::memcpy(buf, &m, sizeof m);
  HIP_TRY(hipMemcpyAsync((char*)buf + sizeof m, g.acc, (size_t)m.n_cells * sizeof(ndt::CellAcc3), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return NDT_OK;
}

int32_t ndt3d_load_map(ndt3d_handle* h, const void* buf, size_t bytes) {
  using namespace ndt;
  if (!h || !buf || bytes < sizeof(ndt_map_header)) return NDT_ERR_INVALID_ARG;
  ndt_map_header m;
  std::memcpy(&m, buf, sizeof m);
  { const int32_t cs = check_map_header(m, bytes, 3, (uint32_t)sizeof(CellAcc3), h->prm.cell_size); if (cs != NDT_OK) return cs; }
  if (m.ngrid != 1) { set_error("a 3D map has one grid"); return NDT_ERR_INVALID_ARG; }
  { const int32_t cs = check_map_cells3((const char*)buf + sizeof m, (size_t)m.n_cells); if (cs != NDT_OK) return cs; }
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  h->has_target = false;
  Grid3Dev& g = h->grid;
  const double c = h->prm.cell_size;
  g.cell = c; g.inv_c = (float)(1.0 / c);
  g.ox = m.origin[0][0]; g.oy = m.origin[0][1]; g.oz = m.origin[0][2];
  g.W = m.width; g.H = m.height; g.D = m.depth; g.pad = 0;
  g.fix_scale = std::ldexp(1.0, kFixShift) / c;
  const size_t ncell = (size_t)m.n_cells;
  if (ncell > h->cell_capacity) {
    void* old[] = {g.rec, g.acc};
    for (void* p : old) if (p) (void)hipFree(p);
    g.rec = nullptr; g.acc = nullptr; h->cell_capacity = 0;
    const size_t want = ncell + ncell / 8;
    HIP_TRY(hipMalloc((void**)&g.rec, 4 * want * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&g.acc, want * sizeof(CellAcc3)));
    h->cell_capacity = want;
  }
  HIP_TRY(hipMemcpyAsync(g.acc, (const char*)buf + sizeof m, ncell * sizeof(CellAcc3), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, ndt::kCountInts * sizeof(int), h->stream));
  hipLaunchKernelGGL(k_finalise3, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, g,
                     h->prm.min_points, h->prm.eig_ratio, h->d_counters);
  HIP_TRY(hipGetLastError());
  int* hc = (int*)h->h_small;
  HIP_TRY(hipMemcpyAsync(hc, h->d_counters, ndt::kCountInts * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));      // buf is free on return
  int n_valid_sum = 0, n_over_sum = 0;
  sum_count_shards(hc, &n_valid_sum, &n_over_sum);
  h->n_valid = n_valid_sum;
  if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
  h->has_target = true;
  return upload_static3(h);
}

}  // extern "C"
